// capi.cpp -- version / error plumbing of the C ABI (include/tmix.h)
#include "common.h"
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void tmix_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local TmixProf g_prof = {nullptr, 0, 0, 0};
TmixProf& tmix_prof_state() { return g_prof; }

extern "C" int tmix_prof_begin(uint64_t* slots, int capacity, int detail) {
    if (!slots || capacity < 1 || (((uintptr_t)slots) & 7)) TMIX_FAIL(TMIX_EINVAL, "prof_begin: need an 8-byte aligned device buffer of capacity >= 1 slots");
    g_prof.buf = (unsigned long long*)slots; g_prof.cap = capacity; g_prof.next = 0; g_prof.detail = detail ? 1 : 0;
    return TMIX_OK;
}
extern "C" int tmix_prof_end(void) {
    const int used = g_prof.next;
    g_prof.buf = nullptr; g_prof.cap = 0; g_prof.next = 0; g_prof.detail = 0;
    return used;
}

static thread_local const char* g_pf_ptr = nullptr;
static thread_local long long g_pf_bytes = 0;
void tmix_prefetch_take(const char** ptr, long long* bytes) {
    *ptr = g_pf_bytes > 0 ? g_pf_ptr : nullptr; *bytes = g_pf_bytes > 0 ? g_pf_bytes : 0;
    g_pf_ptr = nullptr; g_pf_bytes = 0;
}
extern "C" int tmix_gemm_prefetch_next(const void* next_weights, int64_t bytes, void* /*stream*/) {
    if (bytes > (1ll << 31)) TMIX_FAIL(TMIX_ESHAPE, "gemm_prefetch_next: %lld bytes (at most 2 GiB)", (long long)bytes);
    g_pf_ptr = (next_weights && bytes > 0) ? (const char*)next_weights : nullptr;
    g_pf_bytes = g_pf_ptr ? bytes : 0;
    return TMIX_OK;
}

static const char* const g_env_names[TMIX_ENV_COUNT] = {"TMIX_GN_NO_SMALL", "TMIX_ATTN_NO_SPLIT", "TMIX_ATTN_GENERAL", "TMIX_NARROW_EPILOGUE"};
static int g_env[TMIX_ENV_COUNT];
static bool g_env_read = false;     // (racing first users read the same environment and store the same values)
extern "C" void tmix_env_refresh(void) {
    for (int i = 0; i < TMIX_ENV_COUNT; ++i) g_env[i] = getenv(g_env_names[i]) != nullptr;
    g_env_read = true;
}
bool tmix_env(int which) {
    if (!g_env_read) tmix_env_refresh();
    return g_env[which] != 0;
}

extern "C" int tmix_version(void) { return TMIX_VERSION; }
extern "C" const char* tmix_last_error_string(void) { return g_err; }

extern "C" int tmix_check_device(void) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) TMIX_FAIL((int)e, "hipGetDevice: %s", hipGetErrorString(e));
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) TMIX_FAIL((int)e, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) TMIX_FAIL(TMIX_EARCH, "device arch %s is not gfx950", p.gcnArchName);
    return TMIX_OK;
}
