import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if getattr(config.option, "durations", None) is None:      # the 15 slowest tests in every log (the GPU suite ran 1017 s of the driver's 1200 s limit)
        config.option.durations = 15
        config.option.durations_min = 1.0
    # The fp32 ORACLES (oracle/*.py: stock torch ops) run their convolutions on the GPU box too.  Through MIOpen every new fp32 conv configuration is searched /
    # JIT-compiled on first use on a fresh box (no kernel cache travels): tens of seconds per full-size oracle (the 1024^2 VAE decode: 80 s of a 590 s suite,
    # 1017 s on the driver's box in round 5).  With the MIOpen path off torch runs them as unfold + rocBLAS GEMM: deterministic cost, nothing compiled.  The PRODUCT
    # never reaches torch convolutions (its kernels are the HIP library's), so this only changes what the checker costs.  TMIX_TEST_MIOPEN=1 switches it back on.
    if not os.environ.get("TMIX_TEST_MIOPEN"):
        import torch
        torch.backends.cudnn.enabled = False


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a device AND the built library: skip (do not fail) on a CPU box.  On a GPU box a missing
    libtmix_hip.so still fails loudly -- there is no fallback path to hide behind."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (torch.cuda.is_available() is False)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def sdxl_weights():
    """the real SDXL-base parameter shapes (2.57 B), synthetic values, fp32 on the device, built once for the headline-size tests."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tweediemix_amd import unet as U, weights as Wt
    return Wt.synthetic_state_dict(U.SDXL, seed=1234, device="cuda", dtype=torch.float32)


@pytest.fixture(scope="session")
def sdxl_bundles(sdxl_weights):
    """kind -> (concept state dicts, product UNetWeights, fp32 UNetOracle) on the session's SDXL weights, built once per kind ('lora' merged sets /
    'custom'): the per-concept merges and the kernel-layout conversion of 2.57 B parameters are the same for every full-size test."""
    from trajectory_parity import sdxl_bundle
    cache = {}

    def get(kind):
        if kind not in cache:
            cache[kind] = sdxl_bundle(sdxl_weights, kind)
        return cache[kind]
    return get
