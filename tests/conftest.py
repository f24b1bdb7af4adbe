import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
