"""GPU parity of every C-ABI kernel against a plain torch fp32 reference of the same op
(and, for the fused Tweedie step, against the numpy oracle bit-for-bit).

Tolerances: kernels read bf16 operands and accumulate in fp32; outputs are rounded to bf16 once, so
|err| <= 2 bf16 ulp of the result (rtol 2^-7) plus an absolute term for cancellation.
"""
import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture
def lib_env():
    """set one of the library's environment switches mid-process.  The library reads them once per process (a launch and the query that accounts for it must agree);
    tmix_env_refresh() re-reads them.  Restored on teardown."""
    from tweediemix_amd import lib as L
    saved = {}

    def set_(name, value="1"):
        saved.setdefault(name, os.environ.get(name))
        os.environ[name] = value
        L.load().tmix_env_refresh()
    yield set_
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    L.load().tmix_env_refresh()


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tweediemix_amd import lib, ops as O
    lib.check(lib.load().tmix_check_device(), "tmix_check_device")
    return O


def rnd(*shape, seed=0, scale=1.0, dtype=BF):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def close(out, ref, rtol=2 ** -7, atol_frac=2e-3):
    out = out.float()
    ref = ref.float()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all()
    atol = atol_frac * ref.abs().max().item() + 1e-6
    err = (out - ref).abs()
    bad = err > (atol + rtol * ref.abs())
    assert not bad.any(), f"max err {err.max().item():.4g} (ref max {ref.abs().max().item():.4g}), {int(bad.sum())} bad"


# --------------------------------------------------------------------------- fused tweedie step
@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("mode", ["fusion", "plain", "resample"])
@pytest.mark.parametrize("K,h,w,last", [(3, 16, 16, False), (3, 128, 128, False), (2, 8, 12, True), (5, 5, 7, False)])
def test_tweedie_step_bit_exact(ops, dt, mode, K, h, w, last):
    from oracle import tweedie_oracle as TO
    from tweediemix_amd import lib as L
    rng = np.random.RandomState(K * 100 + h)
    x = rng.randn(1, 4, h, w).astype(np.float32)
    eps = rng.randn(K + 1, 4, h, w).astype(np.float32)
    masks = (rng.rand(K, 1, h, w) > 0.6).astype(np.float32)
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    eps_t = torch.from_numpy(eps).to(tdt).cuda()
    eps_r = eps_t.float().cpu().numpy()                 # what the kernel actually reads
    at, an, g = np.float32(0.2345), np.float32(0.3456), 0.8
    lowp = np.float16 if dt == "f16" else None
    if mode == "fusion":
        ref, ref0 = TO.fused_fusion_step(x, eps_r, masks, g, at, an, last, lowp)
        m = L.STEP_FUSION
    elif mode == "plain":
        ref, ref0 = TO.fused_plain_step(x, eps_r[:2], g, at, an, last, lowp)
        m = L.STEP_PLAIN
    else:
        ref = TO.fused_resample_down(x, eps_r, K, g, at, an, lowp)
        ref0 = None
        m = L.STEP_RESAMPLE
        last = False
    xt = torch.from_numpy(x).cuda()
    x0 = torch.empty_like(xt)
    out = ops.fused_tweedie_step(xt, eps_t, torch.from_numpy(masks).cuda(), m, K, g, at, an, last, out_x0=x0)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), ref), np.abs(out.cpu().numpy() - ref).max()
    if ref0 is not None:
        assert np.array_equal(x0.cpu().numpy(), ref0)


@pytest.mark.parametrize("dt", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("mode", ["fusion", "plain", "resample"])
@pytest.mark.parametrize("seeds,per_seed_masks,last", [(1, False, False), (3, True, False), (2, False, True)])
def test_tweedie_step_device_params_multi_seed_in_place(ops, dt, mode, seeds, per_seed_masks, last):
    """tmix_fused_tweedie_step_dev (coefficients from device memory, all co-batched seeds in one launch, latent updated IN
    PLACE -- the form the captured whole-step graph uses) against the numpy oracle bit for bit, seed by seed."""
    from oracle import tweedie_oracle as TO
    from tweediemix_amd import lib as L, ops as O
    lib = L.load()
    K, h, w = 3, 12, 20
    rows = 2 if mode == "plain" else K + 1
    rng = np.random.RandomState(7 + seeds)
    x = rng.randn(seeds, 4, h, w).astype(np.float32)
    eps = rng.randn(seeds * rows, 4, h, w).astype(np.float32)
    masks = (rng.rand(seeds if per_seed_masks else 1, K, 1, h, w) > 0.5).astype(np.float32)
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dt]
    eps_t = torch.from_numpy(eps).to(tdt).cuda()
    eps_r = eps_t.float().cpu().numpy()
    at, an, g = np.float32(0.4111), np.float32(0.5222), 0.8
    lowp = np.float16 if dt == "f16" else None
    m = {"fusion": L.STEP_FUSION, "plain": L.STEP_PLAIN, "resample": L.STEP_RESAMPLE}[mode]
    if mode == "resample":
        last = False
    sa, s1, san, s1n = O.step_coeffs(at, an)
    prm = torch.tensor([781.0, sa, s1, san, s1n, 1.0 if last else 0.0, g, 0.0], dtype=torch.float32).cuda()
    xt = torch.from_numpy(x).cuda()
    x0 = torch.full_like(xt, float("nan"))
    mt = torch.from_numpy(masks).cuda()
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.tmix_fused_tweedie_step_dev(xt.data_ptr(), eps_t.data_ptr(), {"f32": L.F32, "f16": L.F16, "bf16": L.BF16}[dt],
                                         mt.data_ptr(), K * h * w if per_seed_masks else 0, xt.data_ptr(), x0.data_ptr(), K, 4, h * w, m,
                                         rows, seeds, prm.data_ptr(), st)
    L.check(rc, "tmix_fused_tweedie_step_dev")
    torch.cuda.synchronize()
    for sd in range(seeds):
        e = eps_r[sd * rows:(sd + 1) * rows]
        mk = masks[sd if per_seed_masks else 0]
        if mode == "fusion":
            ref, ref0 = TO.fused_fusion_step(x[sd:sd + 1], e, mk, g, at, an, last, lowp)
        elif mode == "plain":
            ref, ref0 = TO.fused_plain_step(x[sd:sd + 1], e, g, at, an, last, lowp)
        else:
            ref, ref0 = TO.fused_resample_down(x[sd:sd + 1], e, K, g, at, an, lowp), None
        assert np.array_equal(xt[sd:sd + 1].cpu().numpy(), ref), (sd, np.abs(xt[sd:sd + 1].cpu().numpy() - ref).max())
        if ref0 is not None:
            assert np.array_equal(x0[sd:sd + 1].cpu().numpy(), ref0)


def test_step_prologue_broadcasts_latent_and_timestep(ops):
    from tweediemix_amd import lib as L
    lib = L.load()
    seeds, rows, h, w = 3, 4, 16, 24
    x = torch.randn(seeds, 4, h, w, device="cuda")
    lat = torch.zeros(seeds * rows, 4, h, w, device="cuda")
    t_dev = torch.zeros(seeds * rows, device="cuda")
    prm = torch.tensor([421.0, 0, 0, 0, 0, 0, 0, 0], dtype=torch.float32, device="cuda")
    L.check(lib.tmix_step_prologue(x.data_ptr(), lat.data_ptr(), t_dev.data_ptr(), prm.data_ptr(), seeds, rows, 4 * h * w,
                                   torch.cuda.current_stream().cuda_stream), "tmix_step_prologue")
    torch.cuda.synchronize()
    assert torch.equal(lat.view(seeds, rows, 4, h, w), x.unsqueeze(1).expand(-1, rows, -1, -1, -1))
    assert torch.equal(t_dev, torch.full_like(t_dev, 421.0))


def test_prof_hook_times_launches_also_inside_a_graph(ops):
    """tmix_prof_begin / tmix_prof_end: every instrumented launch takes one slot, the slots are filled when the work runs
    (eagerly or as a hipGraph replay), and a launch outside the bracket is untouched."""
    from tweediemix_amd import lib as L
    lib = L.load()
    a, w_ = rnd(1024, 1280, seed=1), rnd(1280, 1280, seed=2, scale=0.03)
    x = rnd(2, 32 * 32, 320, seed=3)
    gam, bet = torch.ones(320, device="cuda"), torch.zeros(320, device="cuda")
    slots = torch.zeros(4, 8, dtype=torch.int64, device="cuda")
    slots[:, 0] = -1
    init = slots.clone()

    def work():
        ops.gemm(a, w_)
        ops.groupnorm(x, gam, bet, 32, 1e-5, silu=True)

    work()
    torch.cuda.synchronize()
    L.check(lib.tmix_prof_begin(slots.data_ptr(), 4, 0), "tmix_prof_begin")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        work()
    assert lib.tmix_prof_end() == 2
    assert torch.equal(slots, init)                      # capture does not execute
    work()                                               # outside the bracket: no slot, nothing written
    torch.cuda.synchronize()
    assert torch.equal(slots, init)
    for _ in range(2):
        slots.copy_(init)
        g.replay()
        torch.cuda.synchronize()
        s = slots.cpu().numpy().astype("uint64")
        for k in range(2):
            dur_us = (int(s[k, 1]) - int(s[k, 0])) * 0.01
            assert 0.5 < dur_us < 5000.0, (k, s[k])
        assert int(s[1, 0]) >= int(s[0, 0])              # the norm starts after the GEMM started (same stream)
        assert (s[2:] == init.cpu().numpy().astype("uint64")[2:]).all()


def test_tweedie_step_rejects_bad_args(ops):
    from tweediemix_amd import lib as L
    x = torch.zeros(1, 4, 8, 8, device="cuda")
    eps = torch.zeros(2, 4, 8, 8, device="cuda")
    with pytest.raises(L.TmixError):
        ops.fused_tweedie_step(x, eps, None, L.STEP_FUSION, 3, 0.8, 0.5, 0.6)      # too few rows
    with pytest.raises(L.TmixError):
        ops.fused_tweedie_step(x, eps, None, L.STEP_PLAIN, 3, 0.8, 0.0, 0.6)       # sqrt(alpha) == 0


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (77, 256, 2048), (200, 320, 320),
                                   (1024, 1280, 640), (130, 132, 192), (512, 512, 64)])
def test_gemm_plain(ops, M, N, K, cfg):
    a, w = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    out = ops.gemm(a, w, tile_cfg=cfg)
    close(out, a.float() @ w.float().T)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23])
def test_gemm_epilogues(ops, cfg):
    M, N, K = 384, 640, 256
    a, w = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=K ** -0.5)
    bias = rnd(N, seed=5, dtype=torch.float32)
    res = rnd(M, N, seed=6)
    rgb = rnd(3, N, seed=7, dtype=torch.float32)
    out = ops.gemm(a, w, bias=bias, residual=res, rowgroup_bias=rgb, rows_per_group=128, tile_cfg=cfg)
    ref = a.float() @ w.float().T + bias + res.float() + rgb.repeat_interleave(128, 0)
    close(out, ref)


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 24])
def test_gemm_geglu(ops, cfg):
    M, C = 200, 128
    a = rnd(M, C, seed=8)
    w = rnd(8 * C, C, seed=9, scale=C ** -0.5)           # torch layout: rows [0,4C) value, [4C,8C) gate
    b = rnd(8 * C, seed=10, dtype=torch.float32)
    from tweediemix_amd.weights import interleave_geglu
    wi, bi = interleave_geglu(w, b)
    out = ops.gemm(a, wi, bias=bi, geglu=True, tile_cfg=cfg)
    y = a.float() @ w.float().T + b
    ref = y[:, :4 * C] * F.gelu(y[:, 4 * C:])
    close(out, ref)


def test_gelu_epilogue_is_the_exact_erf_form_also_in_the_tail(ops):
    """gelu_erf_f (common.h): Phi(-|x|) = exp2 of a degree-7 fit -- relative accuracy must hold where gelu(x) is tiny (x -> -5),
    which the rational erf approximation it replaced did not give.  x sweeps [-5.5, 5.5] through an identity weight matrix."""
    n = 64
    xs = torch.linspace(-5.5, 5.5, 4096 * n, dtype=torch.float64).reshape(4096, n)
    a = xs.to(torch.bfloat16).cuda()
    out = ops.gemm(a, torch.eye(n, dtype=torch.bfloat16).cuda(), act="gelu", tile_cfg=1).float().cpu().double()
    x = a.float().cpu().double()
    ref = x * 0.5 * torch.erfc(-x / 2 ** 0.5)                 # exact, no cancellation in the negative tail
    rel = ((out - ref).abs() / ref.abs().clamp_min(1e-30))[x != 0]
    assert rel.max().item() < 2 ** -8 * 1.01 + 1e-5, rel.max().item()        # bf16 rounding of the result + 1e-5 of the function itself


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
def test_gemm_batched_weights_and_transposed_out(ops, cfg):
    Bz, M, C = 3, 100, 128
    a = rnd(Bz, M, C, seed=11)
    w = rnd(Bz, 3 * C, C, seed=12, scale=C ** -0.5)      # one weight set per batch row (concept routing)
    ldvt = 104
    vt = torch.zeros(Bz, C, ldvt, device="cuda", dtype=BF)
    qk = ops.gemm(a, w, out_t=vt, n_trans_begin=2 * C, tile_cfg=cfg)
    ref = torch.einsum("bmk,bnk->bmn", a.float(), w.float())
    close(qk, ref[:, :, :2 * C])
    close(vt[:, :, :M], ref[:, :, 2 * C:].transpose(1, 2))
    assert (vt[:, :, M:] == 0).all()
    # shared weights, strided (column-sliced) A
    big = rnd(Bz, M, 2 * C, seed=13)
    out = ops.gemm(big[:, :, C:], w[0])
    close(out, big[:, :, C:].float() @ w[0].float().T)


@pytest.mark.parametrize("cfg", [1, 2, 4, 6, 7, 13, 14, 16, 17, 18, 21, 22])
def test_wide_epilogue_is_bit_identical_to_the_narrow_one(ops, cfg, monkeypatch, lib_env):
    """the LDS-staged 16-byte stores (default) against the accumulator-layout 8-byte stores (TMIX_NARROW_EPILOGUE=1, the
    fallback for unaligned rows): same values, same operation order per element => identical C, GEGLU output and V^T; the
    row statistics are summed in a different (fixed) order, so they agree to fp32 rounding only."""
    M, N, K = 520, 640, 256                                   # M not a tile multiple: edge rows on both paths
    a, w = rnd(M, K, seed=3), rnd(N, K, seed=4, scale=K ** -0.5)
    bias, res = rnd(N, seed=5, dtype=torch.float32), rnd(M, N, seed=6)
    rgb = rnd(5, N, seed=7, dtype=torch.float32)
    a3, w3 = rnd(2, 264, K, seed=8), rnd(2, 768, K, seed=9, scale=K ** -0.5)
    from tweediemix_amd.weights import interleave_geglu
    wi, bi = interleave_geglu(rnd(1024, K, seed=10, scale=K ** -0.5), rnd(1024, seed=11, dtype=torch.float32))

    def run():
        st = torch.zeros(ops.stats_parts(N, cfg), M, 2, device="cuda")
        c = ops.gemm(a, w, bias=bias, residual=res, rowgroup_bias=rgb, rows_per_group=104, tile_cfg=cfg)
        c2 = ops.gemm(a, w, bias=bias, residual=res, row_stats_out=st, tile_cfg=cfg)
        g = ops.gemm(a, wi, bias=bi, geglu=True, tile_cfg=cfg)
        vt = torch.zeros(2, 256, 264, device="cuda", dtype=BF)
        qk = ops.gemm(a3, w3, out_t=vt, n_trans_begin=512, tile_cfg=cfg)
        f32 = torch.zeros(M, N, device="cuda")
        ops.gemm(a, w, out=None, out_f32=f32, tile_cfg=cfg)
        torch.cuda.synchronize()
        return c, c2, g, vt, qk, f32, st
    wide = run()
    lib_env("TMIX_NARROW_EPILOGUE")
    narrow = run()
    for x, y, name in zip(wide[:6], narrow[:6], ("C", "C+stats", "GEGLU", "Vt", "QK", "f32")):
        assert torch.equal(x, y), name
    torch.testing.assert_close(wide[6], narrow[6], rtol=1e-5, atol=1e-4)
    s = wide[6].sum(0)                                        # and the statistics are those of the stored rows
    torch.testing.assert_close(s[:, 0], wide[1].float().sum(1), rtol=1e-4, atol=1e-3)
    torch.testing.assert_close(s[:, 1], (wide[1].float() ** 2).sum(1), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B,H,Sq,Skv", [(1, 1, 64, 1), (2, 3, 1000, 77), (1, 2, 70, 96), (2, 2, 300, 33), (1, 20, 1024, 77),
                                          (4, 20, 1024, 77), (2, 9, 4096, 77)])     # the last two: five-wave workgroups (the very last with surplus waves)
def test_short_key_attention_kernel_matches_the_general_one(ops, B, H, Sq, Skv, monkeypatch, lib_env):
    """attn_small_kernel (K / V^T register-resident, exact softmax; Skv <= 96) vs torch and vs the tiled flash kernel
    (TMIX_ATTN_GENERAL=1) on the same inputs, incl. ragged query counts and a single key."""
    Cc = H * 64
    q, k, v = rnd(B, Sq, Cc, seed=50), rnd(B, Skv, Cc, seed=51), rnd(B, Skv, Cc, seed=52)
    ld = (Skv + 7) // 8 * 8
    vt = torch.zeros(B, Cc, ld, device="cuda", dtype=BF)
    vt[:, :, :Skv] = v.transpose(1, 2)
    small = ops.attention(q, k, vt, H, Skv, 0.125)
    lib_env("TMIX_ATTN_GENERAL")
    general = ops.attention(q, k, vt, H, Skv, 0.125)

    def heads(t):
        return t.float().reshape(B, -1, H, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), scale=0.125).transpose(1, 2).reshape(B, Sq, Cc)
    close(small, ref, rtol=2 ** -6, atol_frac=4e-3)
    close(small, general.float(), rtol=2 ** -6, atol_frac=4e-3)


def test_gemm_rejects_bad_shapes(ops):
    from tweediemix_amd import lib as L
    with pytest.raises(L.TmixError):
        ops.gemm(rnd(64, 96), rnd(64, 96))                # K % 64
    with pytest.raises(L.TmixError):
        ops.gemm(rnd(64, 64), rnd(66, 64))                # N % 4


# --------------------------------------------------------------------------- conv
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 20])
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 16, 16, 64, 128), (1, 10, 6, 128, 68), (3, 8, 8, 320, 320)])
def test_conv3x3(ops, mode, B, H, W, Cin, Cout, cfg):
    x = rnd(B, H, W, Cin, seed=20)
    w = rnd(Cout, 3, 3, Cin, seed=21, scale=(9 * Cin) ** -0.5)
    bias = rnd(Cout, seed=22, dtype=torch.float32)
    temb = rnd(B, Cout, seed=23, dtype=torch.float32)
    Ho, Wo = ops.conv_out_hw(H, W, mode)
    res = rnd(B, Ho, Wo, Cout, seed=24)
    out = ops.conv3x3(x, w, bias=bias, batch_bias=temb, residual=res, mode=mode, tile_cfg=cfg)
    xn = x.float().permute(0, 3, 1, 2)
    wn = w.float().permute(0, 3, 1, 2)
    if mode == 2:
        xn = F.interpolate(xn, scale_factor=2.0, mode="nearest")
    ref = F.conv2d(xn, wn, bias, stride=2 if mode == 1 else 1, padding=1)
    ref = ref + temb[:, :, None, None] + res.float().permute(0, 3, 1, 2)
    close(out, ref.permute(0, 2, 3, 1))


def test_conv_in_out(ops):
    B, H, W = 2, 12, 20
    x = rnd(B, 4, H, W, seed=30, dtype=torch.float32)
    w = rnd(64, 3, 3, 4, seed=31, scale=1 / 6, dtype=torch.float32)
    b = rnd(64, seed=32, dtype=torch.float32)
    y = ops.conv_in(x, w, b)
    ref = F.conv2d(x, w.permute(0, 3, 1, 2), b, padding=1).permute(0, 2, 3, 1)
    close(y, ref)
    xi = rnd(B, H, W, 96, seed=33)
    wo = rnd(4, 3, 3, 96, seed=34, scale=(9 * 96) ** -0.5)
    bo = rnd(4, seed=35, dtype=torch.float32)
    yo = ops.conv_out(xi, wo, bo)
    ref = F.conv2d(xi.float().permute(0, 3, 1, 2), wo.float().permute(0, 3, 1, 2), bo, padding=1)
    assert yo.dtype == torch.float32
    torch.testing.assert_close(yo, ref, rtol=1e-4, atol=1e-4)


# --------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,H,Sq,Skv", [(2, 2, 256, 256), (1, 3, 64, 77), (2, 1, 200, 16), (1, 2, 1024, 1024), (4, 5, 128, 77)])
def test_attention(ops, B, H, Sq, Skv):
    Cc = H * 64
    q = rnd(B, Sq, Cc, seed=40)
    k = rnd(B, Skv, Cc, seed=41)
    v = rnd(B, Skv, Cc, seed=42)
    ld = (Skv + 7) // 8 * 8
    vt = torch.zeros(B, Cc, ld, device="cuda", dtype=BF)
    vt[:, :, :Skv] = v.transpose(1, 2)
    out = ops.attention(q, k, vt, H, Skv, 0.125)

    def heads(t):
        return t.float().reshape(B, -1, H, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(heads(q), heads(k), heads(v), scale=0.125)
    close(out, ref.transpose(1, 2).reshape(B, Sq, Cc), rtol=2 ** -6, atol_frac=4e-3)


@pytest.mark.parametrize("B,H,Sq,Skv", [(2, 2, 256, 256), (1, 3, 64, 77), (1, 2, 1024, 1024), (4, 5, 128, 77), (2, 1, 200, 80)])
def test_attention_output_as_e4m3_with_mx_block_scales(ops, B, H, Sq, Skv):
    """tmix_attn_fwd_f8 (both kernels: the pipelined one and the short-key one): bytes and scales equal a torch MX quantiser applied to the bf16
    tensor tmix_attn_fwd writes -- so the out-projection behind it reads exactly what a quantiser launch would have produced."""
    Cc = H * 64
    q, k, v = rnd(B, Sq, Cc, seed=40), rnd(B, Skv, Cc, seed=41) * 3, rnd(B, Skv, Cc, seed=42)
    v[:, :, 5] *= 40.0                                         # one loud channel: its 32-column block gets its own scale
    ld = (Skv + 7) // 8 * 8
    vt = torch.zeros(B, Cc, ld, device="cuda", dtype=BF)
    vt[:, :, :Skv] = v.transpose(1, 2)
    out = ops.attention(q, k, vt, H, Skv, 0.125)
    cp = ops.F8Copy(B * Sq, Cc, "cuda")
    cp.buf.fill_(0x5a)
    ops.attention(q, k, vt, H, Skv, 0.125, f8_out=cp)
    torch.cuda.synchronize()
    qq, ss, _deq = _mx_quantize(out.float().view(B * Sq, Cc))
    assert torch.equal(cp.scales, ss)
    same = (cp.q == qq) | (((cp.q & 0x7f) == 0) & ((qq & 0x7f) == 0))
    assert same.all(), int((~same).sum())


def test_attention_online_softmax_rescale(ops):
    """a late key with a huge score forces the running max to jump in the last tile (guide rule 26)."""
    B, H, Sq, Skv = 1, 1, 128, 256
    q = rnd(B, Sq, 64, seed=43)
    k = rnd(B, Skv, 64, seed=44)
    v = rnd(B, Skv, 64, seed=45)
    k[0, 200] = q[0, 5] * 4.0
    k[0, 70] = q[0, 9] * 3.0
    vt = v.transpose(1, 2).contiguous()
    out = ops.attention(q, k, vt, H, Skv, 0.125)
    ref = F.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None], scale=0.125)[:, 0]
    close(out, ref, rtol=2 ** -6, atol_frac=4e-3)


def test_attention_strided_qkv(ops):
    """q/k read straight out of a fused [B,S,2C] projection buffer (row stride 2C)."""
    B, H, S = 2, 2, 192
    Cc = H * 64
    qk = rnd(B, S, 2 * Cc, seed=46)
    v = rnd(B, S, Cc, seed=47)
    vt = v.transpose(1, 2).contiguous()
    out = ops.attention(qk[:, :, :Cc], qk[:, :, Cc:], vt, H, S, 0.125)

    def heads(t):
        return t.float().reshape(B, -1, H, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(heads(qk[:, :, :Cc]), heads(qk[:, :, Cc:]), heads(v), scale=0.125)
    close(out, ref.transpose(1, 2).reshape(B, S, Cc), rtol=2 ** -6, atol_frac=4e-3)


@pytest.mark.parametrize("B,H,S,parts", [(4, 20, 1024, 4), (4, 10, 4096, 2), (8, 20, 1024, 2), (1, 20, 4096, 4)])
def test_attention_key_split_tail(ops, B, H, S, parts):
    """tmix_attn_fwd_ws: the items of the partly filled last round (SDXL attn1 at 1024^2: 640 items at S = 1024, 1280 at S = 4096 on 512 slots) are cut
    into key ranges that meet in the workspace.  Against the fp32 reference, against the unsplit launch (fp32 merge rounding only: a bf16 ulp), bit-stable
    from launch to launch (the merge order is the range order, not the arrival order), ticket counters back at zero; rows with one dominant late key
    (the ranges' maxima differ by far) and the e4m3 output form included."""
    from tweediemix_amd import lib as L
    Cc = H * 64
    l = L.load()
    need = l.tmix_attn_split_ws_bytes(B, H, S, S)
    items, slots = B * H * (S // 128), 512
    assert need == 4096 + (items % slots) * parts * 4 * 9 * 64 * 16
    qk = rnd(B, S, 2 * Cc, seed=48)
    v = rnd(B, S, Cc, seed=49)
    hq, hk = (H - 1) * 64, Cc + (H - 1) * 64                  # the last head of the last batch row is in the split tail: a loud key in its LAST
    qk[B - 1, S - 5, hk:hk + 64] = qk[B - 1, 7, hq:hq + 64] * 4.0       # range for query 7, one in the first range for query 900
    qk[B - 1, 3, hk:hk + 64] = qk[B - 1, 900, hq:hq + 64] * 4.0
    vt = v.transpose(1, 2).contiguous()
    q, k = qk[:, :, :Cc], qk[:, :, Cc:]
    base = ops.attention(q, k, vt, H, S, 0.125)
    ws = ops.attention_split_ws(B, H, S, S, "cuda")
    assert ws is not None and ws.numel() == need
    out = ops.attention(q, k, vt, H, S, 0.125, ws=ws)
    out2 = ops.attention(q, k, vt, H, S, 0.125, ws=ws)
    torch.cuda.synchronize()
    assert torch.equal(out, out2)
    assert int(ws[:4096].view(torch.int32).abs().sum()) == 0
    n_full = items - items % slots                              # whole items are untouched by the split
    rows_full = n_full // (S // 128) // H                       # batch rows made of whole items only
    if rows_full:
        assert torch.equal(out[:rows_full], base[:rows_full])
    assert not torch.equal(out, base)                           # ... and the tail really took the other path
    d = (out.float() - base.float()).abs()
    assert float(d.max()) <= 2 ** -7 * float(base.float().abs().max())

    def heads(t):
        return t.float().reshape(t.shape[0], -1, H, 64).transpose(1, 2)
    for b0 in range(0, B, 2):                                   # (fp32 scores of two batch rows at a time)
        sl = slice(b0, min(B, b0 + 2))
        ref = F.scaled_dot_product_attention(heads(q[sl]), heads(k[sl]), heads(v[sl]), scale=0.125)
        close(out[sl], ref.transpose(1, 2).reshape(-1, S, Cc), rtol=2 ** -6, atol_frac=4e-3)
    # e4m3 output: the bytes a quantiser makes of the bf16 tensor of the SAME (split) launch
    cp = ops.F8Copy(B * S, Cc, "cuda")
    ops.attention(q, k, vt, H, S, 0.125, f8_out=cp, ws=ws)
    torch.cuda.synchronize()
    qq, ss, _deq = _mx_quantize(out.float().view(B * S, Cc))
    assert torch.equal(cp.scales, ss)
    same = (cp.q == qq) | (((cp.q & 0x7f) == 0) & ((qq & 0x7f) == 0))
    assert same.all(), int((~same).sum())


def test_attention_key_split_applies_to_partly_filled_last_rounds_only(ops):
    from tweediemix_amd import lib as L
    l = L.load()
    for shape in [(2, 20, 1024, 1024), (16, 20, 1024, 1024), (4, 20, 1024, 77), (4, 20, 1024, 1000), (1, 1, 128, 128), (4, 20, 1024, 64)]:
        assert l.tmix_attn_split_ws_bytes(*shape) == 0, shape
        assert ops.attention_split_ws(*shape, "cuda") is None
    # a workspace that is too small is refused, not overrun
    B, H, S = 4, 20, 1024
    qk, v = rnd(B, S, 2 * H * 64, seed=48), rnd(B, S, H * 64, seed=49)
    small = torch.zeros(8192, device="cuda", dtype=torch.uint8)
    with pytest.raises(RuntimeError):
        ops.attention(qk[:, :, :H * 64], qk[:, :, H * 64:], v.transpose(1, 2).contiguous(), H, S, 0.125, ws=small)
    # a shape that does not split ignores the workspace
    o1 = ops.attention(qk[:2, :, :H * 64], qk[:2, :, H * 64:], v[:2].transpose(1, 2).contiguous(), H, S, 0.125, ws=small)
    o0 = ops.attention(qk[:2, :, :H * 64], qk[:2, :, H * 64:], v[:2].transpose(1, 2).contiguous(), H, S, 0.125)
    assert torch.equal(o0, o1)


# --------------------------------------------------------------------------- norms / small ops
@pytest.mark.parametrize("B,HW,C1,C2,silu", [(32, 336, 1280, 0, True), (32, 84, 1280, 0, False), (32, 336, 1280, 640, True), (8, 1344, 320, 0, True),
                                             (3, 84, 640, 320, True), (2, 1024, 320, 320, False), (1, 100, 64, 0, True)])
def test_groupnorm_of_small_images_in_one_launch(ops, monkeypatch, B, HW, C1, C2, silu, lib_env):
    """tmix_groupnorm_nhwc on images whose (HW x channels of a few groups) slice is small runs statistics + apply in ONE launch, one workgroup per (image, group set)
    (the video UNet's 336- / 84-pixel frames).  Against torch, against the three-launch form (TMIX_GN_NO_SMALL=1: same statistics up to fp32 summation order), and
    batch-independent: an image's bits do not depend on how many images share the launch."""
    x1 = rnd(B, HW, C1, seed=54) * 1.5 + 0.5
    x2 = rnd(B, HW, C2, seed=55) * 2 - 0.25 if C2 else None
    Cc = C1 + C2
    g, b = rnd(Cc, seed=56, dtype=torch.float32), rnd(Cc, seed=57, dtype=torch.float32)
    y = ops.groupnorm(x1, g, b, 32, 1e-5, silu, x2=x2)
    xin = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], -1)
    ref = F.group_norm(xin.transpose(1, 2), 32, g, b, 1e-5)
    if silu:
        ref = F.silu(ref)
    close(y, ref.transpose(1, 2), rtol=2 ** -6, atol_frac=4e-3)
    y1 = ops.groupnorm(x1[:1].contiguous(), g, b, 32, 1e-5, silu, x2=None if x2 is None else x2[:1].contiguous())
    assert torch.equal(y1, y[:1])
    lib_env("TMIX_GN_NO_SMALL")
    y3 = ops.groupnorm(x1, g, b, 32, 1e-5, silu, x2=x2)
    torch.cuda.synchronize()
    d = (y.float() - y3.float()).abs()
    assert float(d.max()) <= 2 ** -6 * float(y3.float().abs().max())


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(2, 256, 320, 0, True), (1, 100, 64, 0, False), (2, 64, 640, 320, True),
                                             (1, 1024, 1280, 1280, True), (4, 16, 32, 0, False)])
def test_groupnorm(ops, B, HW, C1, C2, silu):
    x1 = rnd(B, HW, C1, seed=50) + 0.5
    x2 = rnd(B, HW, C2, seed=51) * 2 if C2 else None
    Cc = C1 + C2
    g = rnd(Cc, seed=52, dtype=torch.float32)
    b = rnd(Cc, seed=53, dtype=torch.float32)
    y = ops.groupnorm(x1, g, b, 32, 1e-5, silu, x2=x2)
    xin = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], -1)
    ref = F.group_norm(xin.transpose(1, 2), 32, g, b, 1e-5)
    if silu:
        ref = F.silu(ref)
    close(y, ref.transpose(1, 2))


def _colstats_ref(out):
    """[rows/32, 2, C]: column sums and sums of squares of the STORED bf16 values over each 32-row block (fp64 reference)"""
    y = out.reshape(-1, out.shape[-1]).double()
    blk = y.reshape(y.shape[0] // 32, 32, y.shape[1])
    return torch.stack([blk.sum(1), (blk * blk).sum(1)], 1)


def _colstats_close(cs, out):
    ref = _colstats_ref(out)
    assert cs.shape == ref.shape and torch.isfinite(cs).all()
    err = (cs.double() - ref).abs()
    tol = 1e-5 * ref.abs() + 1e-5 * ref[:, 1:].sqrt().max() + 1e-6          # fp32 sums of 32 values
    assert (err <= tol).all(), f"max err {err.max().item():.4g} at {ref.abs().max().item():.4g}"


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5, 7, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
@pytest.mark.parametrize("M,N,K,res", [(256, 320, 128, True), (1024, 1280, 64, True), (96, 200, 64, False), (4096, 640, 64, False)])
def test_gemm_leaves_groupnorm_column_statistics(ops, cfg, M, N, K, res):
    """col_stats_out: the producer-side GroupNorm statistics (straight-line and generic staged epilogues, every tiling, ragged tile edges)"""
    a, w = rnd(M, K, seed=61), rnd(N, K, seed=62, scale=K ** -0.5)
    bias = rnd(N, seed=63, dtype=torch.float32)
    r = rnd(M, N, seed=64) if res else None
    cs = ops.colstats_buf(M, N, "cuda")
    cs.fill_(float("nan"))                                     # every element is written
    out = ops.gemm(a, w, bias=bias, residual=r, tile_cfg=cfg, col_stats_out=cs)
    assert torch.equal(out, ops.gemm(a, w, bias=bias, residual=r, tile_cfg=cfg))       # C itself is untouched by the option
    _colstats_close(cs, out)


@pytest.mark.parametrize("cfg", [1, 2, 4, 5, 7, 12, 13, 14, 15, 20])
@pytest.mark.parametrize("mode,B,H,W,Cin,Cout,res", [(0, 2, 16, 16, 64, 320, True), (1, 1, 16, 16, 128, 64, False), (2, 3, 8, 8, 64, 136, False),
                                                      (0, 2, 32, 32, 64, 640, True)])
def test_conv_leaves_groupnorm_column_statistics(ops, cfg, mode, B, H, W, Cin, Cout, res):
    x = rnd(B, H, W, Cin, seed=65)
    w = rnd(Cout, 3, 3, Cin, seed=66, scale=(9 * Cin) ** -0.5)
    bias = rnd(Cout, seed=67, dtype=torch.float32)
    temb = rnd(B, Cout, seed=68, dtype=torch.float32)
    Ho, Wo = ops.conv_out_hw(H, W, mode)
    r = rnd(B, Ho, Wo, Cout, seed=69) if res else None
    cs = ops.colstats_buf(B * Ho * Wo, Cout, "cuda")
    cs.fill_(float("nan"))
    out = ops.conv3x3(x, w, bias=bias, batch_bias=temb, residual=r, mode=mode, tile_cfg=cfg, col_stats_out=cs)
    assert torch.equal(out, ops.conv3x3(x, w, bias=bias, batch_bias=temb, residual=r, mode=mode, tile_cfg=cfg))
    _colstats_close(cs, out)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4, 5, 7, 12, 13, 14, 15, 20])
@pytest.mark.parametrize("B,H,W,Cin,Cout,c1,c2,cs", [(2, 16, 16, 64, 128, 64, 0, False), (1, 32, 32, 128, 320, 192, 128, True), (3, 8, 8, 64, 72, 64, 64, False),
                                                     (2, 16, 16, 320, 320, 640, 320, True)])
def test_conv_with_shortcut_taps(ops, cfg, B, H, W, Cin, Cout, c1, c2, cs):
    """ResnetBlock2D's conv2(h) + conv_shortcut(cat[x1, x2]) in ONE launch: the 1x1 taps ride behind the nine 3x3 taps (tmix_conv_desc.S1 / S2)"""
    h = rnd(B, H, W, Cin, seed=80)
    w = rnd(Cout, 3, 3, Cin, seed=81, scale=(9 * Cin) ** -0.5)
    x1 = rnd(B, H, W, c1, seed=82)
    x2 = rnd(B, H, W, c2, seed=83) if c2 else None
    wsc = rnd(Cout, c1 + c2, seed=84, scale=(c1 + c2) ** -0.5)
    bias = rnd(Cout, seed=85, dtype=torch.float32)
    temb = rnd(B, Cout, seed=86, dtype=torch.float32)
    csb = ops.colstats_buf(B * H * W, Cout, "cuda") if cs else None
    out = ops.conv3x3(h, ops.shortcut_weight(w, wsc), bias=bias, batch_bias=temb, tile_cfg=cfg, shortcut=(x1, x2), col_stats_out=csb)
    xin = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], -1)
    ref = F.conv2d(h.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    ref = ref + xin @ wsc.float().T + temb[:, None, None, :]
    close(out, ref)
    if cs:
        _colstats_close(csb, out)


# --------------------------------------------------------------------------- tiling 26 (gemm_convh.hip): the input halo patch resident in LDS
@pytest.mark.parametrize("B,H,W,Cin,Cout,extra", [(1, 4, 32, 64, 160, ""), (2, 8, 32, 128, 320, "trc"), (1, 32, 32, 1280, 1280, "rc"), (3, 12, 64, 192, 160, "t"),
                                                  (2, 64, 64, 320, 640, "tc"), (4, 128, 128, 320, 320, "trc"), (1, 16, 96, 64, 480, "r")])
def test_conv3x3_with_the_halo_patch_in_lds(ops, B, H, W, Cin, Cout, extra):
    """TMIX_TILE_CONV_HALO: 4 x 32 pixel tiles, channel-chunk-major K loop, the nine taps as shifted fragment reads of ONE (4 + 2) x (32 + 2) patch per 64-channel chunk.
    Against F.conv2d (image borders = the patch's zero padding, tile borders inside the image = real neighbours, several images, 1 .. 20 chunks), with the time-embedding
    row, the residual and the GroupNorm column statistics of the ResnetBlock2D launches; and against tiling 20 (same products, tap-major order: equal to fp32 rounding)."""
    from tweediemix_amd import lib as L
    x = rnd(B, H, W, Cin, seed=500)
    w = rnd(Cout, 3, 3, Cin, seed=501, scale=(9 * Cin) ** -0.5)
    bias = rnd(Cout, seed=502, dtype=torch.float32)
    temb = rnd(B, Cout, seed=503, dtype=torch.float32) if "t" in extra else None
    res = rnd(B, H, W, Cout, seed=504) if "r" in extra else None
    cs = ops.colstats_buf(B * H * W, Cout, "cuda") if "c" in extra else None
    if cs is not None:
        cs.fill_(float("nan"))
    out = ops.conv3x3(x, w, bias=bias, batch_bias=temb, residual=res, tile_cfg=L.TILE_CONV_HALO, col_stats_out=cs)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    if temb is not None:
        ref = ref + temb[:, None, None, :]
    if res is not None:
        ref = ref + res.float()
    close(out, ref)
    other = ops.conv3x3(x, w, bias=bias, batch_bias=temb, residual=res, tile_cfg=20)
    d = (out.float() - other.float()).abs().max().item()
    assert d <= 2 ** -7 * max(1.0, ref.abs().max().item()), d            # one bf16 ulp of the largest output: the two differ by the fp32 summation order only
    if cs is not None:
        _colstats_close(cs, out)


def test_halo_conv_falls_back_for_launches_it_does_not_carry(ops):
    """stride 2, nearest x2, a width that is not a multiple of 32 or of 160 output channels (with or without shortcut taps) run as tiling 20: the same bits as asking for it"""
    from tweediemix_amd import lib as L
    x = rnd(2, 16, 16, 64, seed=510)
    w = rnd(160, 3, 3, 64, seed=511, scale=(9 * 64) ** -0.5)
    for mode in (0, 1, 2):                                  # W = 16: not a multiple of 32
        assert torch.equal(ops.conv3x3(x, w, mode=mode, tile_cfg=L.TILE_CONV_HALO), ops.conv3x3(x, w, mode=mode, tile_cfg=20))
    x2 = rnd(1, 8, 32, 64, seed=512)
    w2 = rnd(128, 3, 3, 64, seed=513, scale=(9 * 64) ** -0.5)      # Cout = 128
    assert torch.equal(ops.conv3x3(x2, w2, tile_cfg=L.TILE_CONV_HALO), ops.conv3x3(x2, w2, tile_cfg=20))
    wsc = rnd(160, 64, seed=514, scale=64 ** -0.5)
    x1 = rnd(2, 16, 16, 64, seed=515)                              # shortcut taps on a 16-wide image: not its geometry either
    assert torch.equal(ops.conv3x3(x, ops.shortcut_weight(w, wsc), tile_cfg=L.TILE_CONV_HALO, shortcut=(x1, None)),
                       ops.conv3x3(x, ops.shortcut_weight(w, wsc), tile_cfg=20, shortcut=(x1, None)))


@pytest.mark.parametrize("B,H,W,Cin,Cout,c1,c2,cs", [(1, 4, 32, 64, 160, 64, 0, False), (1, 32, 32, 128, 320, 192, 128, True), (2, 8, 64, 64, 160, 64, 64, False),
                                                     (1, 32, 32, 640, 1280, 1280, 640, True), (4, 64, 64, 640, 640, 320, 0, True), (2, 16, 32, 320, 320, 128, 192, False)])
def test_halo_conv_with_shortcut_taps(ops, B, H, W, Cin, Cout, c1, c2, cs):
    """ResnetBlock2D's conv2(h) + conv_shortcut(cat[x1, x2]) in ONE launch of the halo-patch kernel: behind the nine-tap chunks the K loop walks the shortcut tensors'
    64-channel chunks as dense A tiles through the three-slot ring laid over the patch buffers (1 .. 30 shortcut K-tiles, last patch in either buffer)."""
    from tweediemix_amd import lib as L
    h = rnd(B, H, W, Cin, seed=520)
    w = rnd(Cout, 3, 3, Cin, seed=521, scale=(9 * Cin) ** -0.5)
    x1 = rnd(B, H, W, c1, seed=522)
    x2 = rnd(B, H, W, c2, seed=523) if c2 else None
    wsc = rnd(Cout, c1 + c2, seed=524, scale=(c1 + c2) ** -0.5)
    bias = rnd(Cout, seed=525, dtype=torch.float32)
    temb = rnd(B, Cout, seed=526, dtype=torch.float32)
    csb = ops.colstats_buf(B * H * W, Cout, "cuda") if cs else None
    out = ops.conv3x3(h, ops.shortcut_weight(w, wsc), bias=bias, batch_bias=temb, tile_cfg=L.TILE_CONV_HALO, shortcut=(x1, x2), col_stats_out=csb)
    xin = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], -1)
    ref = F.conv2d(h.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    ref = ref + xin @ wsc.float().T + temb[:, None, None, :]
    close(out, ref)
    if cs:
        _colstats_close(csb, out)


def test_column_statistics_reject_what_they_cannot_serve(ops):
    from tweediemix_amd import lib as L
    a, w = rnd(128, 64), rnd(128, 64)
    with pytest.raises(L.TmixError):                          # GEGLU epilogue
        ops.gemm(a, w, geglu=True, col_stats_out=torch.empty(4, 2, 128, device="cuda"))
    with pytest.raises(L.TmixError):                          # transposed region
        ops.gemm(a, w, out_t=torch.empty(128, 128, device="cuda", dtype=BF), n_trans_begin=0, col_stats_out=torch.empty(4, 2, 128, device="cuda"))
    with pytest.raises(L.TmixError):                          # shortcut taps over a channel count that is not a multiple of the K-tile
        ops.conv3x3(rnd(1, 8, 8, 64), ops.shortcut_weight(rnd(64, 3, 3, 64), rnd(64, 32)), shortcut=(rnd(1, 8, 8, 32), None))
    with pytest.raises(L.TmixError):                          # HW not a multiple of 32
        ops.groupnorm(rnd(1, 48, 64), rnd(64, dtype=torch.float32), rnd(64, dtype=torch.float32), colstats=(torch.empty(1, 2, 64, device="cuda"), None))


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(2, 256, 320, 0, True), (1, 1024, 1280, 0, False), (2, 64, 640, 320, True),
                                             (1, 1024, 1280, 640, True), (4, 4096, 320, 0, True), (2, 32, 32, 0, False)])
def test_groupnorm_from_producer_partials(ops, B, HW, C1, C2, silu):
    """tmix_groupnorm_nhwc_pre: the statistics come from column partials (here: exact ones), incl. groups that straddle the two sources (1920 / 32 = 60)"""
    x1 = rnd(B, HW, C1, seed=70) + 0.5
    x2 = rnd(B, HW, C2, seed=71) * 2 if C2 else None
    Cc = C1 + C2
    g = rnd(Cc, seed=72, dtype=torch.float32)
    b = rnd(Cc, seed=73, dtype=torch.float32)
    cs1 = _colstats_ref(x1).float().contiguous()
    cs2 = _colstats_ref(x2).float().contiguous() if C2 else None
    y = ops.groupnorm(x1, g, b, 32, 1e-5, silu, x2=x2, colstats=(cs1, cs2))
    xin = x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], -1)
    ref = F.group_norm(xin.transpose(1, 2), 32, g, b, 1e-5)
    if silu:
        ref = F.silu(ref)
    close(y, ref.transpose(1, 2))
    y0 = ops.groupnorm(x1, g, b, 32, 1e-5, silu, x2=x2)       # the three-launch form: same values up to the summation order of the statistics
    assert (y.float() - y0.float()).abs().max().item() <= 2 ** -6 * ref.abs().max().item()
    if C2:                                                    # ONE concatenated tensor, statistics still in two pieces (the up-blocks' case): bit-equal
        yc = ops.groupnorm(xin.to(BF).contiguous(), g, b, 32, 1e-5, silu, colstats=(cs1, cs2))
        assert torch.equal(yc, y)


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(2, 256, 640, 0, True), (1, 1024, 1280, 640, True), (2, 64, 640, 320, False), (1, 4096, 640, 0, True)])
def test_groupnorm_output_as_e4m3_with_row_major_mx_scales(ops, B, HW, C1, C2, silu):
    """tmix_groupnorm_nhwc_pre_f8: bytes and scales equal a torch MX quantiser applied to the bf16 tensor the plain kernel writes; the scales in the
    ROW-major form [pixels][C / 32] tmix_conv3x3_nhwc_fp8 gathers."""
    x1 = rnd(B, HW, C1, seed=70) + 0.5
    x1[:, :, 7] *= 60.0                                         # a loud channel: its block gets its own scale
    x2 = rnd(B, HW, C2, seed=71) * 2 if C2 else None
    Cc = C1 + C2
    g, b = rnd(Cc, seed=72, dtype=torch.float32), rnd(Cc, seed=73, dtype=torch.float32)
    cs = (_colstats_ref(x1).float().contiguous(), _colstats_ref(x2).float().contiguous() if C2 else None)
    y = ops.groupnorm(x1, g, b, 32, 1e-5, silu, x2=x2, colstats=cs)
    y8 = torch.full((B, HW, Cc), 0x5a, device="cuda", dtype=torch.uint8)
    s8 = torch.full((B * HW, Cc // 32), 0x5a, device="cuda", dtype=torch.uint8)
    ops.groupnorm(x1, g, b, 32, 1e-5, silu, x2=x2, colstats=cs, f8_out=(y8, s8))
    torch.cuda.synchronize()
    q, sc, _deq = _mx_quantize(y.float().view(B * HW, Cc))       # scales come back k-block major [C/32][rows]
    assert torch.equal(s8, sc.t().contiguous())
    same = (y8.view(B * HW, Cc) == q) | (((y8.view(B * HW, Cc) & 0x7f) == 0) & ((q & 0x7f) == 0))
    assert same.all(), int((~same).sum())


@pytest.mark.parametrize("cfg", [12, 20])
@pytest.mark.parametrize("mode,B,H,W,Cin,Cout,extra", [(0, 2, 16, 16, 256, 320, "tr"), (0, 1, 32, 32, 1280, 1280, "r"), (1, 2, 16, 16, 128, 160, ""), (2, 1, 8, 12, 256, 256, "t"),
                                                        (0, 4, 8, 8, 640, 640, "c")])
def test_conv3x3_fp8_equals_the_conv_of_the_dequantised_operands(ops, cfg, mode, B, H, W, Cin, Cout, extra):
    """tmix_conv3x3_nhwc_fp8 (e4m3 input with row-major MX block scales, e4m3 weights with one scale per output channel; a K-tile = 128 channels of one
    tap, the scales gathered per tap with a 4-byte LDS-DMA per row) against F.conv2d of exactly the values it reads; padding, stride 2, nearest x2,
    time-embedding bias, residual and the GroupNorm column statistics as in the bf16 kernel."""
    x = rnd(B, H, W, Cin, seed=300, dtype=torch.float32)
    x[..., 40:72] *= 23.0
    w = rnd(Cout, 3, 3, Cin, seed=301, scale=(9 * Cin) ** -0.5)
    q, sc, xd = _mx_quantize(x.view(B * H * W, Cin))
    sx = sc.t().contiguous()
    w8, sw = ops.quantize_fp8_rows(w.view(Cout, 9 * Cin))
    wd = ops.dequantize_fp8_rows(w8, sw).view(Cout, 3, 3, Cin)
    bias = rnd(Cout, seed=302, dtype=torch.float32)
    Ho, Wo = ops.conv_out_hw(H, W, mode)
    temb = rnd(B, Cout, seed=303, dtype=torch.float32) if "t" in extra else None
    res = rnd(B, Ho, Wo, Cout, seed=304) if "r" in extra else None
    cs = torch.full((B * Ho * Wo // 32, 2, Cout), float("nan"), device="cuda") if "c" in extra else None
    out = ops.conv3x3_fp8(q.view(B, H, W, Cin), sx, w8.view(Cout, 3, 3, Cin), sw, bias=bias, batch_bias=temb, residual=res, mode=mode, tile_cfg=cfg, col_stats_out=cs)
    xin = xd.view(B, H, W, Cin).permute(0, 3, 1, 2)
    if mode == 2:
        xin = F.interpolate(xin, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, wd.permute(0, 3, 1, 2), bias, stride=2 if mode == 1 else 1, padding=1).permute(0, 2, 3, 1)
    if temb is not None:
        ref = ref + temb[:, None, None, :]
    if res is not None:
        ref = ref + res.float()
    close(out, ref, rtol=2 ** -6, atol_frac=4e-3)
    if cs is not None:
        _colstats_close(cs, out)


def test_conv_groupnorm_chain_through_the_partials(ops):
    """conv -> GroupNorm + SiLU as the UNet plan issues it: conv with col_stats_out, then tmix_groupnorm_nhwc_pre"""
    B, H, W, Cin, Cout = 2, 32, 32, 64, 320
    x = rnd(B, H, W, Cin, seed=75)
    w = rnd(Cout, 3, 3, Cin, seed=76, scale=(9 * Cin) ** -0.5)
    g, b = rnd(Cout, seed=77, dtype=torch.float32), rnd(Cout, seed=78, dtype=torch.float32)
    cs = ops.colstats_buf(B * H * W, Cout, "cuda")
    h = ops.conv3x3(x, w, col_stats_out=cs, tile_cfg=12).view(B, H * W, Cout)
    y = ops.groupnorm(h, g, b, 32, 1e-5, True, colstats=(cs, None))
    ref = F.silu(F.group_norm(h.float().transpose(1, 2), 32, g, b, 1e-5)).transpose(1, 2)
    close(y, ref)


@pytest.mark.parametrize("M", [4, 16, 37])
def test_linear_small_sections_matches_separate_launches(ops, M):
    """tmix_linear_small_sections: stacked weight matrices sharing the input, each section leaving as its own dense [M, width]
    (M > 16: co-batched seeds, rows go out 16 per launch)."""
    K = 1280
    widths = [320, 640, 1280, 320]
    x = torch.randn(M, K, device="cuda")
    ws = [rnd(n, K, seed=70 + i, scale=K ** -0.5) for i, n in enumerate(widths)]
    bs = [rnd(n, seed=80 + i, dtype=torch.float32) for i, n in enumerate(widths)]
    st = torch.tensor([0] + list(torch.tensor(widths).cumsum(0)), device="cuda", dtype=torch.int32)
    flat = ops.linear_small_sections(x, torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous(), st, act_in=True)
    off = 0
    for w, b_ in zip(ws, bs):
        n = w.shape[0]
        sep = ops.linear_small(x, w, b_, act_in=True)
        assert torch.equal(flat[off * M:(off + n) * M].view(M, n), sep)
        close(sep, F.silu(x) @ w.float().t() + b_)
        off += n


@pytest.mark.parametrize("rows,Cc", [(77, 640), (1024, 1280), (5, 64), (3, 2048)])
def test_layernorm(ops, rows, Cc):
    x = rnd(rows, Cc, seed=54) * 3 + 1
    g = rnd(Cc, seed=55, dtype=torch.float32)
    b = rnd(Cc, seed=56, dtype=torch.float32)
    close(ops.layernorm(x, g, b, 1e-5), F.layer_norm(x.float(), (Cc,), g, b, 1e-5))


def test_concat_and_embedding_and_linear_small(ops):
    a, b = rnd(3, 50, 64, seed=57), rnd(3, 50, 128, seed=58)
    assert torch.equal(ops.concat_channels(a, b), torch.cat([a, b], -1))
    vals = torch.tensor([981.0, 1.0, 1024.0, 0.0, 500.0], device="cuda")
    for dim in (320, 256):
        e = ops.timestep_embedding(vals, dim)
        half = dim // 2
        freq = torch.exp(-np.log(10000.0) * torch.arange(half, device="cuda", dtype=torch.float32) / half)
        arg = vals[:, None] * freq[None]
        ref = torch.cat([torch.cos(arg), torch.sin(arg)], -1)
        torch.testing.assert_close(e, ref, rtol=0, atol=2e-4)
    for M in (1, 4, 6):
        x = rnd(M, 320, seed=59, dtype=torch.float32)
        w = rnd(1280, 320, seed=60, scale=320 ** -0.5)
        bias = rnd(1280, seed=61, dtype=torch.float32)
        add = rnd(M, 1280, seed=62, dtype=torch.float32)
        y = ops.linear_small(x, w, bias, add, act_in=True, act_out=True)
        ref = F.silu(F.silu(x) @ w.float().T + bias + add)
        torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cfg", [1, 2, 4, 7, 8, 9, 10, 13, 14, 15, 20, 21, 22, 23])
def test_gemm_fused_layernorm_pair(ops, cfg):
    """producer GEMM emits row statistics of what it stored, consumer GEMM applies LayerNorm algebraically:
    together == Linear2(LayerNorm(Linear1(a) + res)) of diffusers' BasicTransformerBlock."""
    from tweediemix_amd import lib as L
    from tweediemix_amd.weights import fold_layernorm, interleave_geglu
    import ctypes as C
    M, C1, N2 = 300, 256, 384
    a = rnd(M, C1, seed=70)
    w1 = rnd(C1, C1, seed=71, scale=C1 ** -0.5)
    b1 = rnd(C1, seed=72, dtype=torch.float32)
    res = rnd(M, C1, seed=73) * 2 + 0.7                      # non-zero row means
    gamma = rnd(C1, seed=74, dtype=torch.float32) * 0.2 + 1
    beta = rnd(C1, seed=75, dtype=torch.float32) * 0.3
    w2 = rnd(N2, C1, seed=76, scale=C1 ** -0.5)
    b2 = rnd(N2, seed=77, dtype=torch.float32)
    parts = ops.stats_parts(C1, cfg)
    stats = torch.full((parts, M, 2), float("nan"), device="cuda")       # every slot must be written, none accumulated
    h = ops.gemm(a, w1, bias=b1, residual=res, row_stats_out=stats, tile_cfg=cfg)
    hf = h.float()
    torch.testing.assert_close(stats[:, :, 0].sum(0), hf.sum(-1), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(stats[:, :, 1].sum(0), (hf ** 2).sum(-1), rtol=1e-5, atol=1e-3)
    wp, cs, t = fold_layernorm(w2, gamma, beta, b2)
    y = ops.gemm(h, wp, bias=t, ln_stats=stats, ln_colsum=cs, tile_cfg=cfg)
    ref = F.layer_norm(hf, (C1,), gamma, beta, 1e-5) @ w2.float().T + b2
    close(y, ref, rtol=2 ** -6, atol_frac=4e-3)
    # GEGLU consumer and transposed-V consumer
    w3 = rnd(8 * C1, C1, seed=78, scale=C1 ** -0.5)
    b3 = rnd(8 * C1, seed=79, dtype=torch.float32)
    wp3, cs3, t3 = fold_layernorm(w3, gamma, beta, b3)
    wi = interleave_geglu(wp3, None)[0]
    csi, ti = interleave_geglu(cs3[:, None], t3)
    yg = ops.gemm(h, wi, bias=ti, geglu=True, ln_stats=stats, ln_colsum=csi[:, 0].contiguous(), tile_cfg=cfg)
    z = F.layer_norm(hf, (C1,), gamma, beta, 1e-5) @ w3.float().T + b3
    close(yg, z[:, :4 * C1] * F.gelu(z[:, 4 * C1:]), rtol=2 ** -6, atol_frac=4e-3)
    w4 = rnd(3 * C1, C1, seed=80, scale=C1 ** -0.5)
    wp4, cs4, t4 = fold_layernorm(w4, gamma, beta, None)
    vt = torch.zeros(1, C1, 304, device="cuda", dtype=BF)
    qk = ops.gemm(h.unsqueeze(0), wp4, bias=t4, out_t=vt, n_trans_begin=2 * C1, ln_stats=stats, ln_colsum=cs4, tile_cfg=cfg)
    z4 = F.layer_norm(hf, (C1,), gamma, beta, 1e-5) @ w4.float().T
    close(qk[0], z4[:, :2 * C1], rtol=2 ** -6, atol_frac=4e-3)
    close(vt[0, :, :M], z4[:, 2 * C1:].T, rtol=2 ** -6, atol_frac=4e-3)


@pytest.mark.parametrize("r", [1, 10, 100, 300, 1000])
@pytest.mark.parametrize("cfg", [2, 12, 14])
def test_fused_layernorm_with_row_means_far_from_zero(ops, cfg, r):
    """the folded LayerNorm takes var = E[x^2] - mean^2 from fp32 partial sums (gemm_kernel.h ln_reduce) and subtracts mean * colsum(W') from an
    fp32 accumulator: rows with |mean| / std = r lose ~ 2 r^2 * 2^-24 of the variance and r * 2^-24 * sqrt(K) of the products.  Isolated here from
    the bf16 rounding of the hidden state itself (the reference sees the SAME bf16 rows: an identity producer stores them exactly and leaves the
    statistics): exact to the bf16 output rounding up to r = 100 (1.7e-3), 2.4e-3 at 300, 3.2e-2 at 1000 -- the bound asserted is max(4e-3, r^2 * 2^-24), and the reason the plan has no
    two-pass fallback: a bf16 row with r = 100 has already lost 4 of its 8 mantissa bits to the mean before any norm looks at it."""
    from tweediemix_amd.weights import fold_layernorm
    M, C1, N2 = 256, 1280, 320
    z = rnd(M, C1, seed=170).float()
    x = ((z - z.mean(-1, keepdim=True)) / z.std(-1, keepdim=True) + float(r)).to(BF)      # bf16 rows with mean ~ r, std ~ 1 (coarser for large r)
    eye = torch.eye(C1, device="cuda", dtype=BF)
    parts = ops.stats_parts(C1, cfg)
    stats = torch.full((parts, M, 2), float("nan"), device="cuda")
    h = ops.gemm(x, eye, row_stats_out=stats, tile_cfg=cfg)
    assert torch.equal(h, x)
    gamma = rnd(C1, seed=174, dtype=torch.float32) * 0.2 + 1
    beta = rnd(C1, seed=175, dtype=torch.float32) * 0.3
    w2 = rnd(N2, C1, seed=176, scale=C1 ** -0.5)
    b2 = rnd(N2, seed=177, dtype=torch.float32)
    wp, cs, t = fold_layernorm(w2, gamma, beta, b2)
    y = ops.gemm(h, wp, bias=t, ln_stats=stats, ln_colsum=cs, tile_cfg=cfg).double()
    xd = h.double()
    mu, var = xd.mean(-1, keepdim=True), xd.var(-1, unbiased=False, keepdim=True)
    # the reference normalises the same bf16 rows in fp64 and applies the SAME folded bf16 weights (the fold's own rounding is tested above)
    ref = ((xd - mu) / (var + 1e-5).sqrt()) @ wp.double().T + t.double()
    err = ((y - ref).norm() / ref.norm()).item()
    bound = max(4e-3, r * r * 2.0 ** -24)
    print(f"fused LayerNorm, rows with |mean|/std = {r} (actual std {var.sqrt().mean().item():.3g}), tiling {cfg}: rel-L2 {err:.3e} (bound {bound:.3e})")
    assert err <= bound, (err, bound)


@pytest.mark.parametrize("cfg", [0, 1, 2, 7, 13])
def test_conv_temporal_3x1x1(ops, cfg):
    """TMIX_CONV_T3 against F.conv3d with a (3,1,1) kernel and (1,0,0) padding (diffusers TemporalConvLayer):
    x [clips, frames, h*w, C] <-> torch [clips, C, frames, h, w]."""
    from tweediemix_amd import lib as L
    clips, frames, h, w, Ci, Co = 2, 16, 6, 5, 128, 192
    x = rnd(clips, frames, h * w, Ci, seed=90)
    wt = rnd(Co, 3, Ci, seed=91, scale=(3 * Ci) ** -0.5)
    bias = rnd(Co, seed=92, dtype=torch.float32)
    res = rnd(clips, frames, h * w, Co, seed=93)
    y = ops.conv3x3(x, wt, bias=bias, residual=res, mode=L.CONV_T3, tile_cfg=cfg)
    xt = x.float().view(clips, frames, h, w, Ci).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(xt, wt.float().permute(0, 2, 1)[:, :, :, None, None], bias, padding=(1, 0, 0))
    ref = ref.permute(0, 2, 3, 4, 1).reshape(clips, frames, h * w, Co) + res.float()
    close(y, ref, rtol=2 ** -6, atol_frac=4e-3)


@pytest.mark.parametrize("frames", [16, 9])
def test_temporal_attention(ops, frames):
    """tmix_temporal_attn against torch softmax attention over the frame axis ([b*hw, heads, frames, 64])."""
    clips, hw, heads = 2, 37, 5
    C = heads * 64
    qkv = rnd(clips * frames, hw, 3 * C, seed=95)
    out = ops.temporal_attention(qkv, clips, frames, heads)
    x = qkv.float().view(clips, frames, hw, 3, heads, 64).permute(3, 0, 2, 4, 1, 5)      # [3, clips, hw, heads, frames, 64]
    ref = F.scaled_dot_product_attention(x[0], x[1], x[2])                               # over frames
    ref = ref.permute(0, 3, 1, 2, 4).reshape(clips * frames, hw, C)
    close(out, ref, rtol=2 ** -6, atol_frac=4e-3)


@pytest.mark.parametrize("cfg", [1, 2, 4, 6, 7])
def test_gemm_transposed_region_on_a_128_boundary(ops, cfg):
    """q/k/v of a 320-channel layer (I2VGen-XL level 0): the transposed V region starts at column 640, a multiple of 128 but
    not of 256 -- the 256-wide tilings must fall back instead of writing V columns into the row-major output."""
    S, Cc = 200, 320
    x = rnd(2, S, Cc, seed=97)
    wt = rnd(3 * Cc, Cc, seed=98, scale=Cc ** -0.5)
    vt = torch.zeros(2, Cc, 200, device="cuda", dtype=BF)
    guard = torch.full((2, S + 8, 2 * Cc), 7.0, device="cuda", dtype=BF)
    qk = ops.gemm(x, wt, out=guard[:, :S], out_t=vt, n_trans_begin=2 * Cc, tile_cfg=cfg)
    ref = x.float() @ wt.float().T
    close(qk, ref[:, :, :2 * Cc], rtol=2 ** -6, atol_frac=4e-3)
    close(vt, ref[:, :, 2 * Cc:].transpose(1, 2), rtol=2 ** -6, atol_frac=4e-3)
    assert float(guard[:, S:].float().min()) == 7.0 and float(guard[:, S:].float().max()) == 7.0


# --------------------------------------------------------------------------- fp8 (e4m3, per-row E8M0 scales)
@pytest.mark.parametrize("rows,K", [(37, 64), (1024, 1280), (130, 5120)])
def test_fp8_row_quantizer_matches_torch_float8(ops, rows, K):
    """tmix_quantize_fp8_rows: the scale is the smallest power of two that brings the row's largest magnitude under 448, the
    bytes are round-to-nearest-even OCP e4m3 -- bit-equal to torch's float8_e4m3fn conversion of the scaled row."""
    x = rnd(rows, K, seed=rows, scale=3.0)
    x[1] = 0                                              # an all-zero row keeps scale 1
    x[2, 5] = 1000.0                                      # an outlier sets its row's scale
    q, s = ops.quantize_fp8_rows(x)
    torch.cuda.synchronize()
    xf = x.float()
    amax = xf.abs().amax(dim=1)
    e = torch.where(amax > 0, torch.ceil(torch.log2(amax.double() / 448.0)).float(), torch.zeros_like(amax))
    assert torch.equal(s.float() - 127.0, e), (s[:4], e[:4])
    ref = (xf * torch.exp2(-e).unsqueeze(1)).to(torch.float8_e4m3fn).view(torch.uint8)
    same = (q == ref) | ((q & 0x7f) == 0) & ((ref & 0x7f) == 0)        # +0 / -0 both fine
    assert same.all(), int((~same).sum())
    back = ops.dequantize_fp8_rows(q, s)
    assert float((back - xf).abs().max() / xf.abs().max()) < 2 ** -3   # 3 mantissa bits


@pytest.mark.parametrize("tile", [0, 12, 16, 17, 19, 20, 21])       # 12 / 19 / 20 / 21: e4m3 in the lock-step loops (128 x 160 without / with 1, 2, 4 loader waves)
@pytest.mark.parametrize("M,N,K,kw", [(512, 512, 256, {}), (1024, 1280, 1280, {"bias": True, "residual": True}),
                                      (300, 264, 128, {"bias": True}), (2048, 2560, 1280, {"geglu": True}),
                                      (1024, 3840, 1280, {"trans": True, "batch": 2})])
def test_gemm_fp8_equals_fp32_product_of_the_dequantized_operands(ops, tile, M, N, K, kw):
    """tmix_gemm_fp8 (v_mfma_scale_f32_32x32x64_f8f6f4, per-row scales applied by the instruction) against the fp32 matmul of
    exactly the values it reads: products of e4m3 values are exact in fp32, so only the accumulation order and the final bf16
    rounding separate the two.  Also reports what the quantisation itself costs against the unquantised bf16 operands."""
    batch = kw.get("batch", 1)
    a = rnd(batch, M, K, seed=1) if batch > 1 else rnd(M, K, seed=1)
    w = rnd(N, K, seed=2, scale=K ** -0.5)
    a8, sa = ops.quantize_fp8_rows(a)
    w8, sw = ops.quantize_fp8_rows(w)
    ad, wd = ops.dequantize_fp8_rows(a8, sa), ops.dequantize_fp8_rows(w8, sw)
    bias = torch.randn(N, device="cuda") if (kw.get("bias") or kw.get("geglu")) else None
    res = rnd(M, N, seed=3) if kw.get("residual") else None
    ref = ad @ wd.t() + (bias if bias is not None else 0)
    exact = a.float() @ w.float().t() + (bias if bias is not None else 0)
    if kw.get("geglu"):
        from tweediemix_amd.weights import interleave_geglu
        wi, bi = interleave_geglu(w.float(), bias)
        w8, sw = ops.quantize_fp8_rows(wi.to(BF).contiguous())
        wd = ops.dequantize_fp8_rows(w8, sw)
        full = ad @ wd.t() + bi
        # un-interleave: groups of 32 rows = 16 value rows then 16 gate rows
        v = full.view(M, N // 32, 2, 16)
        ref = (v[:, :, 0] * F.gelu(v[:, :, 1])).reshape(M, N // 2)
        out = ops.gemm_fp8(a8, sa, w8, sw, bias=bi.contiguous(), geglu=True, tile_cfg=tile)
        close(out, ref, rtol=2 ** -6)
        return
    if kw.get("trans"):
        ntb = N // 3 * 2
        vt = torch.zeros(batch, N - ntb, M, device="cuda", dtype=BF)
        out = ops.gemm_fp8(a8, sa, w8, sw, bias=bias, out_t=vt, n_trans_begin=ntb, tile_cfg=tile)
        close(out, ref[..., :ntb])
        close(vt, ref[..., ntb:].transpose(1, 2))
        return
    out = ops.gemm_fp8(a8, sa, w8, sw, bias=bias, residual=res, tile_cfg=tile)
    close(out, ref + (res.float() if res is not None else 0))
    err = float((out.float() - (exact + (res.float() if res is not None else 0))).norm() / exact.norm())
    print(f"fp8 GEMM {M}x{N}x{K}: rel-L2 vs the unquantised bf16 product = {err:.3e}")
    assert err < 6e-2


def _mx_quantize(x):
    """torch reference of the MX block form: x fp32 [rows, K] -> (e4m3 bytes [rows, K], E8M0 scales [K/32, rows])."""
    rows, K = x.shape
    xb = x.view(rows, K // 32, 32)
    amax = xb.abs().amax(dim=2)
    e = torch.where(amax > 0, torch.ceil(torch.log2(amax.double() / 448.0)).float(), torch.zeros_like(amax))
    q = (xb * torch.exp2(-e).unsqueeze(2)).to(torch.float8_e4m3fn).view(torch.uint8).view(rows, K)
    return q.contiguous(), (e + 127).to(torch.uint8).t().contiguous(), (q.view(torch.float8_e4m3fn).float().view(rows, K // 32, 32)
                                                                        * torch.exp2(e).unsqueeze(2)).view(rows, K)


@pytest.mark.parametrize("tile", [12, 16, 17, 19, 20, 21])
@pytest.mark.parametrize("M,N,K", [(512, 512, 256), (1024, 1280, 5120), (300, 264, 128)])
def test_gemm_fp8_with_mx_block_scales_on_a(ops, tile, M, N, K):
    """TMIX_F8_A_BLOCK_SCALES: one E8M0 scale per 32 K values of every A row, kept in LDS for the K loop (lane half h of a 64-wide
    slice s applies scale [2 s + h]) -- against the fp32 product of the dequantised operands."""
    a = rnd(M, K, seed=1, dtype=torch.float32)
    a[:, 64:96] *= 37.0                                    # blocks of very different magnitude in one row
    a[5] = 0
    w = rnd(N, K, seed=2, scale=K ** -0.5)
    a8, sa, ad = _mx_quantize(a)
    w8, sw = ops.quantize_fp8_rows(w)
    wd = ops.dequantize_fp8_rows(w8, sw)
    bias = torch.randn(N, device="cuda")
    res = rnd(M, N, seed=3)
    out = ops.gemm_fp8(a8, sa, w8, sw, bias=bias, residual=res, tile_cfg=tile, a_block_scales=True)
    close(out, ad @ wd.t() + bias + res.float())


def test_gemm_fp8_mx_block_scales_batched_and_limits(ops):
    """Batched launches index the scale array [K/32][batch * M] at column b * M + m; K beyond what the tile can keep in LDS beside
    the staging ring, and rows that break the 4-byte pieces of the scale copy, are refused (no silent fallback)."""
    B, M, N, K = 2, 264, 256, 192
    a = rnd(B * M, K, seed=4, dtype=torch.float32)
    a[:, 32:64] *= 19.0
    w = rnd(B, N, K, seed=5, scale=K ** -0.5)
    a8, sa, ad = _mx_quantize(a)
    w8, sw = ops.quantize_fp8_rows(w.view(B * N, K))
    wd = ops.dequantize_fp8_rows(w8, sw).view(B, N, K)
    out = ops.gemm_fp8(a8.view(B, M, K), sa, w8.view(B, N, K), sw.view(B, N), tile_cfg=17, a_block_scales=True)
    close(out, torch.einsum("bmk,bnk->bmn", ad.view(B, M, K), wd))
    from tweediemix_amd.lib import TmixError
    K2 = 32 * 256
    with pytest.raises(TmixError):
        ops.gemm_fp8(torch.zeros(64, K2, device="cuda", dtype=torch.uint8), torch.zeros(K2 // 32, 64, device="cuda", dtype=torch.uint8),
                     torch.zeros(64, K2, device="cuda", dtype=torch.uint8), torch.zeros(64, device="cuda", dtype=torch.uint8), a_block_scales=True)
    with pytest.raises(TmixError):
        ops.gemm_fp8(torch.zeros(66, 64, device="cuda", dtype=torch.uint8), torch.zeros(2, 66, device="cuda", dtype=torch.uint8),
                     torch.zeros(64, 64, device="cuda", dtype=torch.uint8), torch.zeros(64, device="cuda", dtype=torch.uint8), a_block_scales=True)


@pytest.mark.parametrize("tile", [1, 4, 7, 12, 13, 17, 18])
@pytest.mark.parametrize("batch", [1, 2])
def test_gemm_leaves_an_e4m3_copy_of_its_output_for_the_next_gemm(ops, tile, batch):
    """TMIX_F8_COPY_OUT: next to the bf16 rows (bit-identical to a launch without the flag, statistics included) the plain epilogue
    writes what an MX quantiser would produce from them -- e4m3 bytes and one E8M0 scale per 32 columns -- and tmix_gemm_fp8 reads
    that pair as a block-scaled A operand (per batch row for per-row weight sets)."""
    M, N, K = 160, 320, 128
    a = rnd(batch, M, K, seed=20)
    w = rnd(batch, N, K, seed=21, scale=K ** -0.5)
    bias = rnd(N, seed=22, dtype=torch.float32)
    res = rnd(batch, M, N, seed=23)
    parts = ops.stats_parts(N, tile)
    st0 = torch.zeros(parts, batch * M, 2, device="cuda")
    st1 = torch.zeros_like(st0)
    ref = ops.gemm(a, w, bias=bias, residual=res, tile_cfg=tile, row_stats_out=st0)
    cp = ops.F8Copy(batch * M, N, "cuda")
    cp.buf.fill_(0x5a)
    out = ops.gemm(a, w, bias=bias, residual=res, tile_cfg=tile, row_stats_out=st1, f8_copy=cp)
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(st0, st1)
    q, s, deq = _mx_quantize(out.float().view(batch * M, N))
    assert torch.equal(cp.scales, s)
    same = (cp.q == q) | (((cp.q & 0x7f) == 0) & ((q & 0x7f) == 0))
    assert same.all(), int((~same).sum())
    # consumer: the copy as block-scaled A of an fp8 GEMM (weights per batch row)
    w2 = rnd(batch, 256, N, seed=24, scale=N ** -0.5)
    w28, sw2 = ops.quantize_fp8_rows(w2.view(batch * 256, N))
    nxt = ops.gemm_fp8(cp.q.view(batch, M, N), cp.scales, w28.view(batch, 256, N), sw2.view(batch, 256), tile_cfg=17, a_block_scales=True)
    want = torch.einsum("bmk,bnk->bmn", deq.view(batch, M, N), ops.dequantize_fp8_rows(w28, sw2).view(batch, 256, N))
    close(nxt, want)
    from tweediemix_amd.lib import TmixError
    with pytest.raises(TmixError):                                      # rows that do not fill 32-row blocks are refused
        ops.gemm(a[:, :150].contiguous(), w, tile_cfg=tile, f8_copy=ops.F8Copy(batch * 150, N, "cuda"))


@pytest.mark.parametrize("tile", [12, 19, 20, 21])
@pytest.mark.parametrize("batch", [1, 2])
def test_gemm_fp8_lockstep_tilings_leave_copy_and_statistics(ops, tile, batch):
    """the N = 1280 GEMMs of an fp8 plan (attention out-projections, FF2) on e4m3 operands in the 128 x 160 loops: bf16 rows + residual, LayerNorm row
    statistics and the e4m3 + MX-block copy of the stored rows from ONE launch (with loader waves: epilogue family 4) -- each against what the
    phase-offset tiling 17 / a torch MX quantiser produce."""
    M, N, K = 256, 320, 384
    a = rnd(batch * M, K, seed=30, dtype=torch.float32)
    a[:, 32:64] *= 11.0
    w = rnd(batch, N, K, seed=31, scale=K ** -0.5)
    a8, sa, ad = _mx_quantize(a)
    w8, sw = ops.quantize_fp8_rows(w.view(batch * N, K))
    bias = rnd(N, seed=32, dtype=torch.float32)
    res = rnd(batch, M, N, seed=33)
    st = torch.full((ops.stats_parts(N, tile), batch * M, 2), float("nan"), device="cuda")
    cp = ops.F8Copy(batch * M, N, "cuda")
    cp.buf.fill_(0x5a)
    out = ops.gemm_fp8(a8.view(batch, M, K), sa, w8.view(batch, N, K), sw.view(batch, N), bias=bias, residual=res, tile_cfg=tile, a_block_scales=True,
                       row_stats_out=st, f8_copy=cp)
    torch.cuda.synchronize()
    want = torch.einsum("bmk,bnk->bmn", ad.view(batch, M, K), ops.dequantize_fp8_rows(w8, sw).view(batch, N, K)) + bias + res.float()
    close(out, want)
    of = out.float().view(batch * M, N)
    torch.testing.assert_close(st[:, :, 0].sum(0), of.sum(-1), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(st[:, :, 1].sum(0), (of ** 2).sum(-1), rtol=1e-5, atol=1e-3)
    q, s, _deq = _mx_quantize(of)
    assert torch.equal(cp.scales, s)
    same = (cp.q == q) | (((cp.q & 0x7f) == 0) & ((q & 0x7f) == 0))
    assert same.all(), int((~same).sum())


def test_gemm_fp8_lockstep_tilings_refuse_the_e4m3_geglu_output(ops):
    """MX blocks of 32 GEGLU output columns need wave tiles that are multiples of 64 weight rows wide: a 160-wide tile ends inside a block -- refused."""
    from tweediemix_amd.lib import TmixError
    a8 = torch.zeros(256, 256, device="cuda", dtype=torch.uint8); sa = torch.zeros(256, device="cuda", dtype=torch.uint8)
    w8 = torch.zeros(1280, 256, device="cuda", dtype=torch.uint8); sw = torch.zeros(1280, device="cuda", dtype=torch.uint8)
    c8 = torch.zeros(256, 640, device="cuda", dtype=torch.uint8); cs = torch.zeros(20, 256, device="cuda", dtype=torch.uint8)
    with pytest.raises(TmixError):
        ops.gemm_fp8(a8, sa, w8, sw, bias=torch.zeros(1280, device="cuda"), geglu=True, tile_cfg=21, f8_out=(c8, cs))


@pytest.mark.parametrize("tile", [16, 17])
def test_gemm_fp8_geglu_output_as_mx_blocks_feeds_the_next_gemm(ops, tile):
    """TMIX_F8_GEGLU_OUT: the FF up-projection writes value * gelu(gate) as e4m3 with one scale per 32 output columns, exactly
    what a quantiser would produce from the bf16 result, and the down-projection consumes it (A block scales): FF1 -> FF2 without a
    quantiser pass and with half the bytes of the bf16 intermediate."""
    from tweediemix_amd.weights import interleave_geglu
    M, C = 520, 256
    a = rnd(M, C, seed=8)
    w1, b1 = rnd(8 * C, C, seed=9, scale=C ** -0.5), rnd(8 * C, seed=10, dtype=torch.float32)
    wi, bi = interleave_geglu(w1, b1)
    a8, sa = ops.quantize_fp8_rows(a)
    wi8, swi = ops.quantize_fp8_rows(wi.contiguous())
    ref_bf16 = ops.gemm_fp8(a8, sa, wi8, swi, bias=bi, geglu=True, tile_cfg=tile)          # the bf16 form of the same GEMM
    c8 = torch.zeros(M, 4 * C, device="cuda", dtype=torch.uint8)
    cs = torch.zeros(4 * C // 32, M, device="cuda", dtype=torch.uint8)
    ops.gemm_fp8(a8, sa, wi8, swi, bias=bi, geglu=True, tile_cfg=tile, f8_out=(c8, cs))
    torch.cuda.synchronize()
    q, s, deq = _mx_quantize(ref_bf16.float())
    assert torch.equal(cs, s)
    same = (c8 == q) | (((c8 & 0x7f) == 0) & ((q & 0x7f) == 0))
    assert same.all(), int((~same).sum())
    # and the chain: FF2 on the e4m3 intermediate
    w2 = rnd(C, 4 * C, seed=11, scale=(4 * C) ** -0.5)
    w28, sw2 = ops.quantize_fp8_rows(w2)
    out = ops.gemm_fp8(c8, cs, w28, sw2, tile_cfg=tile, a_block_scales=True)
    close(out, deq @ ops.dequantize_fp8_rows(w28, sw2).t())


@pytest.mark.parametrize("cfg", [7, 12, 16, 18, 20, 21])
def test_prefetch_hint_changes_nothing_but_timing(ops, cfg):
    """tmix_gemm_prefetch_next: the launch that consumes the hint touches another tensor while it waits for its own operands --
    its C is bit-identical to the un-hinted launch, the hinted range is only read, and the hint is consumed by ONE launch."""
    import ctypes as C
    from tweediemix_amd import lib as L
    lib = L.load()
    st = torch.cuda.current_stream().cuda_stream
    a, w = rnd(520, 256, seed=3), rnd(640, 256, seed=4, scale=256 ** -0.5)
    nxt = rnd(3000, 1280, seed=5)                      # 7.7 MB of "next weights" (not a multiple of the grid's share)
    keep = nxt.clone()
    ref = ops.gemm(a, w, tile_cfg=cfg).clone()
    L.check(lib.tmix_gemm_prefetch_next(nxt.data_ptr(), nxt.numel() * 2, st))
    got = ops.gemm(a, w, tile_cfg=cfg).clone()
    again = ops.gemm(a, w, tile_cfg=cfg)                # no hint pending any more
    torch.cuda.synchronize()
    assert torch.equal(got, ref) and torch.equal(again, ref) and torch.equal(nxt, keep)
    assert lib.tmix_gemm_prefetch_next(nxt.data_ptr(), 1 << 40, st) < 0          # larger than 2 GiB: refused
    L.check(lib.tmix_gemm_prefetch_next(None, 0, st))                            # clears


@pytest.mark.parametrize("P,ln", [(4, False), (12, False), (4, True), (12, True)])
def test_lora_down_fills_the_pad_columns(ops, P, ln):
    """tmix_lora_down: the row's own concept's down-projections at pad columns [set * P, +P), zeros elsewhere; with a folded
    LayerNorm the value that the GEMM's rstd * (acc - mean * colsum) turns into up(down(LN(x)))."""
    K, nsets, S, B = 320, 4, 72, 5
    sets = torch.tensor([0, 3, 1, 2, 1], dtype=torch.int32).cuda()
    x = rnd(B * S, K, seed=60) * 1.5 + 0.25
    a = torch.full((B * S, K + 64), 7.0, dtype=BF).cuda()                 # stale pad contents must be overwritten
    a[:, :K] = x
    D = rnd(nsets * P, K, seed=61, scale=0.05)
    gamma, beta = rnd(K, seed=62, dtype=torch.float32) * 0.2 + 1.0, rnd(K, seed=63, dtype=torch.float32) * 0.1
    if ln:
        Dp = (D.float() * gamma).to(BF)
        ops.lora_down(a, K, Dp, P, nsets, sets, S, dcolsum=Dp.float().sum(1).contiguous(), dbias=(D.float() @ beta).contiguous())
    else:
        ops.lora_down(a, K, D, P, nsets, sets, S)
    assert torch.equal(a[:, :K], x)
    xf = x.float()
    ref = torch.zeros(B * S, 64, device="cuda")
    for b in range(B):
        s_ = int(sets[b])
        rows = slice(b * S, (b + 1) * S)
        if ln:
            mean = xf[rows].mean(1, keepdim=True)
            sd = (xf[rows].var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()
            Dp32 = (D.float() * gamma).to(BF).float()[s_ * P:(s_ + 1) * P]
            ref[rows, s_ * P:(s_ + 1) * P] = (xf[rows] - mean) @ Dp32.T + (D.float()[s_ * P:(s_ + 1) * P] @ beta) * sd
        else:
            ref[rows, s_ * P:(s_ + 1) * P] = xf[rows] @ D.float()[s_ * P:(s_ + 1) * P].T
    close(a[:, K:], ref)
    mask = ref == 0
    assert (a[:, K:].float()[mask] == 0).all()



@pytest.mark.parametrize("cfg", [0, 2, 4, 7, 12, 13, 14, 16, 17, 19, 20, 21, 23])
def test_gemm_periodic_weight_sets_equal_the_gathered_form(ops, cfg):
    """tmix_gemm_desc.w_period: a batch of seeds x (1 + K) rows against 1 + K stored weight sets (slice b reads set b % P) is the same launch,
    bit for bit, as the one over per-row gathered copies of the weights -- plain, with per-set bias + residual, with the folded LayerNorm
    (per-set ln_colsum), and with a transposed V region."""
    from tweediemix_amd.weights import fold_layernorm
    P, seeds, M, K, N = 4, 3, 200, 256, 640
    Bz = P * seeds
    a = rnd(Bz, M, K, seed=301)
    w = rnd(P, N, K, seed=302, scale=K ** -0.5)
    bias = rnd(P, N, seed=303, dtype=torch.float32)
    res = rnd(Bz, M, N, seed=304)
    idx = torch.arange(Bz, device="cuda") % P
    wg, bg = w[idx].contiguous(), bias[idx].contiguous()
    kw = dict(tile_cfg=cfg) if cfg else {}
    assert torch.equal(ops.gemm(a, w, **kw), ops.gemm(a, wg, **kw))
    got = ops.gemm(a, w, bias=bias, residual=res, **kw)
    assert torch.equal(got, ops.gemm(a, wg, bias=bg, residual=res, **kw))
    close(got, torch.einsum("bmk,bnk->bmn", a.float(), wg.float()) + bg[:, None] + res.float())
    # consumer of a folded LayerNorm: per-set colsum / bias
    gamma, beta = rnd(K, seed=305, dtype=torch.float32) * 0.2 + 1, rnd(K, seed=306, dtype=torch.float32) * 0.3
    fold = [fold_layernorm(w[i], gamma, beta, bias[i]) for i in range(P)]
    wp, cs, t = [torch.stack([f[j] for f in fold]).contiguous() for j in range(3)]
    h = (rnd(Bz, M, K, seed=307) * 2 + 0.5)
    hf = h.float()
    stats = torch.stack([hf.sum(-1), (hf ** 2).sum(-1)], -1).view(1, Bz * M, 2).contiguous()
    y = ops.gemm(h, wp, bias=t, ln_stats=stats, ln_colsum=cs, **kw)
    assert torch.equal(y, ops.gemm(h, wp[idx].contiguous(), bias=t[idx].contiguous(), ln_stats=stats, ln_colsum=cs[idx].contiguous(), **kw))
    close(y, torch.einsum("bmk,bnk->bmn", F.layer_norm(hf, (K,), gamma, beta, 1e-5), wg.float()) + bg[:, None], rtol=2 ** -6, atol_frac=4e-3)
    if cfg != 22:
        w3 = rnd(P, 3 * 128, K, seed=308, scale=K ** -0.5)
        vt1, vt2 = [torch.zeros(Bz, 128, 208, device="cuda", dtype=BF) for _ in range(2)]
        q1 = ops.gemm(a, w3, out_t=vt1, n_trans_begin=256, **kw)
        q2 = ops.gemm(a, w3[idx].contiguous(), out_t=vt2, n_trans_begin=256, **kw)
        assert torch.equal(q1, q2) and torch.equal(vt1, vt2)


@pytest.mark.parametrize("tile", [0, 12, 16, 17, 21])
def test_gemm_fp8_periodic_weight_sets_equal_the_gathered_form(ops, tile):
    P, seeds, M, K, N = 4, 2, 256, 256, 640
    Bz = P * seeds
    a8, sa = ops.quantize_fp8_rows(rnd(Bz, M, K, seed=311))
    w8, sw = ops.quantize_fp8_rows(rnd(P, N, K, seed=312, scale=K ** -0.5))
    idx = torch.arange(Bz, device="cuda") % P
    kw = dict(tile_cfg=tile) if tile else {}
    got = ops.gemm_fp8(a8, sa, w8, sw, **kw)
    assert torch.equal(got, ops.gemm_fp8(a8, sa, w8[idx].contiguous(), sw[idx].contiguous(), **kw))


def test_gemm_rejects_a_period_that_does_not_divide_the_batch(ops):
    import ctypes as C
    from tweediemix_amd import lib as L
    a, w = rnd(6, 64, 64, seed=1), rnd(4, 64, 64, seed=2)
    d = ops.make_gemm_desc(a[:4], w, torch.empty(4, 64, 64, device="cuda", dtype=BF))
    d.batch, d.w_period = 6, 4
    assert L.load().tmix_gemm_bf16(C.byref(d), None) == L.ESHAPE


@pytest.mark.parametrize("period", ["batch", 1])
def test_gemm_period_equal_to_the_batch_is_the_plain_strided_walk(ops, period):
    """tmix.h documents every w_period > 0 that divides the batch as valid.  P == batch ("every slice its own set") made w_groups = 1 and wrapped the
    magic divisor to 1: slices >= 1 indexed A / C far outside the batch.  It is the plain strideW walk; P == 1 is one shared set."""
    import ctypes as C
    from tweediemix_amd import lib as L
    Bz, M, K, N = 4, 128, 128, 256
    a, w = rnd(Bz, M, K, seed=321), rnd(Bz, N, K, seed=322, scale=K ** -0.5)
    want = ops.gemm(a, w)
    guard = torch.full((Bz + 2, M, N), 7.0, device="cuda", dtype=BF)       # a slice in front of and behind the output: must stay untouched
    out = guard[1:1 + Bz]
    d = ops.make_gemm_desc(a, w, out)
    P = Bz if period == "batch" else 1
    d.w_period = P
    if P == 1:
        want = ops.gemm(a, w[:1].expand(Bz, N, K).contiguous())
    L.check(L.load().tmix_gemm_bf16(C.byref(d), torch.cuda.current_stream().cuda_stream), "tmix_gemm_bf16")
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    assert (guard[0] == 7).all() and (guard[-1] == 7).all()


# --------------------------------------------------------------------------- tiling 23 (gemm_w22.hip): 128 x 160 over 2 x 2 waves of 64 x 80
@pytest.mark.parametrize("M,N,K,batch", [(4096, 1280, 1280, 1), (1024, 1280, 1280, 4), (300, 640, 2560, 1), (16384, 640, 640, 1), (77, 160, 64, 2), (129, 320, 192, 3)])
def test_gemm_w22_bias_residual_statistics_and_folded_layernorm(ops, M, N, K, batch):
    """every epilogue flavour the 2 x 2 kernel carries, on the shapes it is shipped for (attention out-projections, attn2 to_q, FF2) and on ragged
    ones (M not a multiple of 128: clamped loads, predicated stores): plain; bias + residual + LayerNorm row statistics (producer side); consumer
    of a folded LayerNorm; one weight set per batch slice.  Against fp32 products of the same bf16 operands."""
    from tweediemix_amd.weights import fold_layernorm
    shp = (batch, M) if batch > 1 else (M,)
    a = rnd(*shp, K, seed=401)
    w = rnd(*((batch,) if batch > 1 else ()), N, K, seed=402, scale=K ** -0.5)
    ref = torch.einsum("...mk,...nk->...mn", a.float(), w.float())
    close(ops.gemm(a, w, tile_cfg=23), ref)
    bias = rnd(*((batch,) if batch > 1 else ()), N, seed=403, dtype=torch.float32)
    res = rnd(*shp, N, seed=404) * 2 + 0.5
    parts = ops.stats_parts(N, 23)
    assert parts == N // 160
    stats = torch.full((parts, batch * M, 2), float("nan"), device="cuda")
    h = ops.gemm(a, w, bias=bias, residual=res, row_stats_out=stats, tile_cfg=23)
    close(h, ref + (bias[:, None] if batch > 1 else bias) + res.float())
    hf = h.float().reshape(batch * M, N)
    torch.testing.assert_close(stats[:, :, 0].sum(0), hf.sum(-1), rtol=1e-5, atol=2e-3)
    torch.testing.assert_close(stats[:, :, 1].sum(0), (hf ** 2).sum(-1), rtol=1e-5, atol=2e-3)
    # the same launch on the four-wave tiling: identical stored values up to the accumulation order of the MFMA shapes
    h21 = ops.gemm(a, w, bias=bias, residual=res, tile_cfg=21)
    assert (h.float() - h21.float()).abs().max().item() <= 2 ** -7 * h21.float().abs().max().item()
    # consumer of the folded LayerNorm over those rows (N of the producer = K of the consumer)
    if N % 64:
        return
    N2 = 320
    gamma, beta = rnd(N, seed=405, dtype=torch.float32) * 0.2 + 1, rnd(N, seed=406, dtype=torch.float32) * 0.3
    w2, b2 = rnd(N2, N, seed=407, scale=N ** -0.5), rnd(N2, seed=408, dtype=torch.float32)
    wp, cs, t = fold_layernorm(w2, gamma, beta, b2)
    y = ops.gemm(h.reshape(batch * M, N), wp, bias=t, ln_stats=stats, ln_colsum=cs, tile_cfg=23)
    close(y, F.layer_norm(hf, (N,), gamma, beta, 1e-5) @ w2.float().T + b2, rtol=2 ** -6, atol_frac=4e-3)


_EXPERIMENTAL = pytest.mark.skipif(not os.environ.get("TMIX_EXPERIMENTAL_TILINGS"),
                                   reason="tilings 24 / 25 live in dev variants only (tools/build_variant.sh x EXPERIMENTAL=1; run with TMIX_LIB=<variant> TMIX_EXPERIMENTAL_TILINGS=1): "
                                          "the shipped library runs those ids as tilings 14 / 23")


@_EXPERIMENTAL
@pytest.mark.parametrize("M,N,K,batch", [(4096, 1280, 5120, 1), (1024, 1280, 1280, 4), (300, 640, 2560, 1), (129, 320, 192, 3), (128, 160, 64, 1)])
def test_gemm_w22_with_l2_prefetcher_wave_equals_tiling_23_bit_for_bit(ops, M, N, K, batch):
    """tiling 25 = tiling 23's math waves behind three DMA loaders and one prefetcher wave: the same bits, whatever the K depth (1 to 80 K-tiles)"""
    shp = (batch, M) if batch > 1 else (M,)
    a = rnd(*shp, K, seed=421)
    w = rnd(*((batch,) if batch > 1 else ()), N, K, seed=422, scale=K ** -0.5)
    bias = rnd(N, seed=423, dtype=torch.float32)
    res = rnd(*shp, N, seed=424)
    s23 = torch.zeros(N // 160, batch * M, 2, device="cuda"); s25 = torch.zeros_like(s23)
    y23 = ops.gemm(a, w, bias=bias, residual=res, row_stats_out=s23, tile_cfg=23)
    y25 = ops.gemm(a, w, bias=bias, residual=res, row_stats_out=s25, tile_cfg=25)
    assert torch.equal(y23, y25) and torch.equal(s23, s25)


def test_gemm_w22_falls_back_for_launches_it_does_not_carry(ops):
    """GEGLU, a transposed region, an activation, a row-group bias or a width that is not a multiple of 160 run as tiling 21 / 12: same results as asking for those"""
    from tweediemix_amd.weights import interleave_geglu
    M, C = 256, 320
    a = rnd(M, C, seed=411)
    w = rnd(8 * C, C, seed=412, scale=C ** -0.5)
    wi, _ = interleave_geglu(w, None)
    assert torch.equal(ops.gemm(a, wi, geglu=True, tile_cfg=23), ops.gemm(a, wi, geglu=True, tile_cfg=21))
    w2 = rnd(384, C, seed=413, scale=C ** -0.5)
    assert torch.equal(ops.gemm(a, w2, tile_cfg=23), ops.gemm(a, w2, tile_cfg=21))
    assert torch.equal(ops.gemm(a, w2[:320], act="gelu", tile_cfg=23), ops.gemm(a, w2[:320], act="gelu", tile_cfg=21))


# --------------------------------------------------------------------------- attn2 in one launch (gemm_qattn.hip)
def _qattn_ref(a, w, bias, k, vt, rows_per_image, scale, ln=None):
    """fp32 reference: q = [LN](a) w^T + bias per batch slice, then softmax(q K^T scale) V per 64-wide head against the image's cached keys"""
    af = a.float()
    if ln is not None:
        af = F.layer_norm(af, (af.shape[-1],), ln[0], ln[1], 1e-5)
    a3 = af if af.dim() == 3 else af.unsqueeze(0)
    w3 = w.float() if w.dim() == 3 else w.float().unsqueeze(0)
    q = torch.einsum("bmk,bnk->bmn", a3, w3.expand(a3.shape[0], -1, -1) if w3.shape[0] == 1 else w3[torch.arange(a3.shape[0]) % w3.shape[0]])
    if bias is not None:
        b2 = bias if bias.dim() == 2 else bias.unsqueeze(0)
        q = q + (b2.expand(a3.shape[0], -1) if b2.shape[0] == 1 else b2[torch.arange(a3.shape[0]) % b2.shape[0]])[:, None]
    q = q.to(BF).float()                                    # the two-launch form stores q in bf16
    Bz, M, N = q.shape
    H, Skv = N // 64, k.shape[1]
    q = q.reshape(Bz * M // rows_per_image, rows_per_image, H, 64).transpose(1, 2)
    kk = k.float().reshape(k.shape[0], Skv, H, 64).transpose(1, 2)
    vv = vt.float()[:, :, :Skv].reshape(vt.shape[0], H, 64, Skv).transpose(2, 3)
    o = F.scaled_dot_product_attention(q, kk, vv, scale=scale)
    return o.transpose(1, 2).reshape(Bz, M, N) if a.dim() == 3 else o.transpose(1, 2).reshape(M, N)


@pytest.mark.parametrize("B,S,C,Skv,routed,ln", [(4, 1024, 1280, 77, True, True), (4, 1024, 1280, 77, False, True), (2, 4096, 640, 77, False, True),
                                                   (1, 128, 320, 80, False, False), (3, 192, 320, 5, True, True), (2, 64, 640, 33, False, False)])
def test_q_projection_and_cross_attention_in_one_launch(ops, B, S, C, Skv, routed, ln):
    """tmix_gemm_q_cross_attn against the fp32 reference AND against the two launches it replaces (tmix_gemm_bf16 -> tmix_attn_fwd) on the shapes of
    the SDXL plan (32 x 32 and 64 x 64 levels, LoRA-routed per-row weights and shared weights, LayerNorm folded into to_q) and on small / ragged ones
    (one to 80 keys, three images, no LayerNorm)."""
    from tweediemix_amd.weights import fold_layernorm
    h = rnd(B, S, C, seed=501) * 1.5 + 0.3
    P = B if routed else 1
    wq = rnd(P, C, C, seed=502, scale=C ** -0.5)
    bq = rnd(P, C, seed=503, dtype=torch.float32) * 0.1
    k = rnd(B, Skv, C, seed=504)
    vt = torch.zeros(B, C, 80, device="cuda", dtype=BF)
    vt[:, :, :Skv] = rnd(B, C, Skv, seed=505)
    scale = 64 ** -0.5
    kw = {}
    lnp = None
    if ln:
        gamma, beta = rnd(C, seed=506, dtype=torch.float32) * 0.2 + 1, rnd(C, seed=507, dtype=torch.float32) * 0.3
        lnp = (gamma, beta)
        fold = [fold_layernorm(wq[i], gamma, beta, bq[i]) for i in range(P)]
        wp, cs, t = [torch.stack([f[j] for f in fold]).contiguous() for j in range(3)]
        hf = h.float()
        stats = torch.stack([hf.sum(-1), (hf ** 2).sum(-1)], -1).view(1, B * S, 2).contiguous()
        kw = dict(ln_stats=stats, ln_colsum=cs if routed else cs[0])
        w_used, b_used = (wp, t) if routed else (wp[0], t[0])
    else:
        w_used, b_used = (wq, bq) if routed else (wq[0], bq[0])
    a = h if routed else h.view(B * S, C)
    got = ops.gemm_q_cross_attn(a, w_used, k, vt, S, scale, bias=b_used, **kw)
    ref = _qattn_ref(a, wq if routed else wq[0], bq if routed else bq[0], k, vt, S, scale, lnp)
    close(got, ref, rtol=2 ** -6, atol_frac=6e-3)
    # the two-launch form on the same operands
    q = ops.gemm(a, w_used, bias=b_used, **kw)
    two = ops.attention(q.view(B, S, C), k, vt, C // 64, Skv, scale)
    d = (got.float().view(B, S, C) - two.float()).abs().max().item()
    assert d <= 2 ** -6 * two.float().abs().max().item() + 1e-3, d


def test_q_cross_attention_rejects_what_it_does_not_carry(ops):
    import ctypes as C_
    from tweediemix_amd import lib as L
    a, w = rnd(128, 320, seed=511), rnd(320, 320, seed=512)
    k, vt = rnd(1, 77, 320, seed=513), torch.zeros(1, 320, 80, device="cuda", dtype=BF)
    out = torch.empty(128, 320, device="cuda", dtype=BF)
    lib = L.load()
    d = ops.make_gemm_desc(a, w, None, residual=out)
    assert lib.tmix_gemm_q_cross_attn(C_.byref(d), *ops.q_cross_attn_args(k, vt, out, 128, 0.125), None) == L.EINVAL
    d = ops.make_gemm_desc(a, rnd(256, 320, seed=514), None)          # four heads: not a multiple of the five-head tile
    assert lib.tmix_gemm_q_cross_attn(C_.byref(d), *ops.q_cross_attn_args(k[:, :, :256], vt[:, :256], out[:, :256], 128, 0.125), None) == L.ESHAPE
    d = ops.make_gemm_desc(a, w, None)
    assert lib.tmix_gemm_q_cross_attn(C_.byref(d), *ops.q_cross_attn_args(k, vt, out, 96, 0.125), None) == L.ESHAPE     # rows_per_image % 64


def test_long_row_quantiser_fallback_uses_the_kernels_scale_arithmetic(ops):
    """rows longer than 8192 (conv weight rows of 9 * 1280) are quantised in torch; the E8M0 scale must be e8m0_for_amax's, bit for bit -- also where
    amax is exactly 7 * 2^n, the boundary on which ceil(log2(amax / 448)) in double and the kernel's fp32 product can disagree."""
    n = 40
    amax = torch.tensor([7.0 * 2.0 ** (k - 20) for k in range(n)] + [448.0, 449.0, 447.0, 1.0, 3.0e-5, 0.0])
    R = amax.numel()
    short = torch.zeros(R, 8192)
    short[:, 5] = amax
    short[:, 77] = -amax * 0.5
    long_ = torch.cat([short, torch.zeros(R, 64)], dim=1)                 # the same rows with 64 zero columns behind them: the torch path
    q1, s1 = ops.quantize_fp8_rows(short.to(BF).cuda())
    q2, s2 = ops.quantize_fp8_rows(long_.to(BF).cuda())
    assert torch.equal(s1, s2)
    assert torch.equal(q1, q2[:, :8192])


# --------------------------------------------------------------------------- tiling 24 (gemm_ff1p.hip): 256 x 320 on persistent workgroups
@_EXPERIMENTAL
@pytest.mark.parametrize("M,N,K,ln", [(4096, 10240, 1280, True), (16384, 5120, 640, True), (2048, 10240, 1280, True), (256, 320, 128, False), (768, 960, 192, True), (512, 20480, 64 * 3, False)])
def test_gemm_persistent_geglu_equals_tiling_14_bit_for_bit(ops, M, N, K, ln):
    """the FF up-projection on persistent workgroups (one per CU walking 1, 2, 4 ... tiles: 512 / 1024 / 256 tiles at the SDXL shapes, and grids smaller
    than the chip): same tile, same MFMA order, same epilogue arithmetic as tiling 14 -> identical bits; and both against the fp32 reference."""
    from tweediemix_amd.weights import fold_layernorm, interleave_geglu
    h = rnd(M, K, seed=601) * 1.3 + 0.4
    w = rnd(N, K, seed=602, scale=K ** -0.5)
    b = rnd(N, seed=603, dtype=torch.float32)
    kw = {}
    if ln:
        gamma, beta = rnd(K, seed=604, dtype=torch.float32) * 0.2 + 1, rnd(K, seed=605, dtype=torch.float32) * 0.3
        wp, cs, t = fold_layernorm(w, gamma, beta, b)
        wi = interleave_geglu(wp, None)[0]
        csi, ti = interleave_geglu(cs[:, None], t)
        hf = h.float()
        stats = torch.stack([hf.sum(-1), (hf ** 2).sum(-1)], -1).view(1, M, 2).contiguous()
        kw = dict(ln_stats=stats, ln_colsum=csi[:, 0].contiguous())
        z = F.layer_norm(hf, (K,), gamma, beta, 1e-5) @ w.float().T + b
    else:
        wi, ti = interleave_geglu(w, b)
        z = h.float() @ w.float().T + b
    y24 = ops.gemm(h, wi, bias=ti, geglu=True, tile_cfg=24, **kw)
    y14 = ops.gemm(h, wi, bias=ti, geglu=True, tile_cfg=14, **kw)
    assert torch.equal(y24, y14)
    close(y24, z[:, :N // 2] * F.gelu(z[:, N // 2:]), rtol=2 ** -6, atol_frac=4e-3)


# --------------------------------------------------------------------------- the fp16 rounding points, by the device's own torch
@pytest.mark.parametrize("mode", ["fusion", "plain", "resample", "last"])
@pytest.mark.parametrize("K,h,w", [(3, 16, 16), (3, 128, 128), (2, 8, 12)])
def test_fp16_step_equals_the_references_statements_evaluated_by_torch_on_the_device(ops, mode, K, h, w):
    """the fp16-eps mode's bit-level claim rested on the oracle's CPU_SCALAR_TENSOR_SEMANTICS switch ("an argument, not a vector").  Here the vector is made on
    the spot: the statements of fusion_sampling.py:376-385 (fusion), :424-430 (plain), :391-402 (first half of a resampling repeat) and :471-472 (t == 1) are evaluated by torch ON THE
    DEVICE -- fp16 noise_pred (what the UNet returns under autocast, :492), fp32 x, python-float guidance scale, `at` / `at_next` 0-dim fp32 CPU tensors exactly as
    `self.scheduler.alphas_cumprod[t]` hands them over (:305-307) -- i.e. with the scalar-tensor promotion and rounding the reference really executes on a GPU, and the fused kernel's
    fp16 mode must reproduce the result (to the reciprocal-vs-division ulp, see below)."""
    from tweediemix_amd import lib as L
    g = torch.Generator().manual_seed(K * 1000 + h)
    x = torch.randn(1, 4, h, w, generator=g).cuda()
    noise_pred = torch.randn(K + 1, 4, h, w, generator=g).half().cuda()
    masks = (torch.rand(K, 1, h, w, generator=g) > 0.6).float().cuda()
    at, at_next = torch.tensor(0.2345), torch.tensor(0.3456)             # 0-dim fp32 CPU tensors (alphas_cumprod lives on the CPU)
    gs = 0.8
    noise_pred_uncond = noise_pred[:1]
    if mode in ("fusion", "last"):
        denoised_tweedie = 0
        for cc in range(K):
            noise_pred_cond = noise_pred[(1 + cc):(2 + cc)]
            noise_pred_concept = noise_pred_uncond + gs * (noise_pred_cond - noise_pred_uncond)
            denoised_tweedie += masks[cc].unsqueeze(0) * ((x - (1 - at).sqrt() * noise_pred_concept) / at.sqrt())
        m = L.STEP_FUSION
    elif mode == "plain":
        noise_pred_cond = noise_pred[1:2]
        npred = noise_pred_uncond + gs * (noise_pred_cond - noise_pred_uncond)
        denoised_tweedie = (x - (1 - at).sqrt() * npred) / at.sqrt()
        m = L.STEP_PLAIN
    else:
        noise_pred_mult = noise_pred[1:2]
        noise_pred_mult = noise_pred_uncond + gs * (noise_pred_mult - noise_pred_uncond)
        denoised_tweedie_mult = (x - (1 - at).sqrt() * noise_pred_mult) / at.sqrt()
        denoised_tweedie = (K - 1) * denoised_tweedie_mult
        for cc in range(K - 1):
            noise_pred_single = noise_pred_uncond + gs * (noise_pred[2 + cc:3 + cc] - noise_pred_uncond)
            denoised_tweedie_single = (x - (1 - at).sqrt() * noise_pred_single) / at.sqrt()
            denoised_tweedie -= denoised_tweedie_single
        m = L.STEP_RESAMPLE
    denoised_latent = at_next.sqrt() * denoised_tweedie + (1 - at_next).sqrt() * noise_pred_uncond
    if mode == "last":
        denoised_latent = denoised_tweedie
    assert denoised_latent.dtype == torch.float32 and denoised_tweedie.dtype == torch.float32
    x0 = torch.empty_like(x)
    out = ops.fused_tweedie_step(x, noise_pred, masks, m, K, gs, np.float32(at.item()), np.float32(at_next.item()), mode == "last", out_x0=x0)
    torch.cuda.synchronize()
    # agreement to a few fp32 ulps, not bit for bit: torch's GPU kernel divides by a CPU-scalar tensor as a multiplication by its reciprocal (the fused kernel and the numpy
    # oracle -- pinned to fixtures the reference's code produced on the CPU -- divide), a 1e-7 effect; a wrong fp16 rounding point would show at 1e-3 of an element
    tol = 4e-6 * denoised_latent.abs().max().item()
    assert (out - denoised_latent).abs().max().item() <= tol, (out - denoised_latent).abs().max().item()
    if mode != "resample":
        assert (x0 - denoised_tweedie).abs().max().item() <= 4e-6 * denoised_tweedie.abs().max().item()
