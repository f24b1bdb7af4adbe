"""GPU parity of the video-sampler pieces (BASELINE config #5): the fused CFG + v-prediction DDIM update
(tmix_vpred_step) and the first-frame feature injection (tmix_frame_inject) against oracle/tweedie_oracle.py and the
vectors of tests/golden/video_step.npz (produced by executing the reference's statements).  fp32: 2e-6 abs;
fp16: bit-exact against the oracle with CUDA scalar semantics (the device the reference runs on), loose against the
torch-CPU fixture, whose coefficients were rounded to fp16 first (the oracle reproduces THAT bit-exactly on the CPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import tweedie_oracle as O

pytestmark = pytest.mark.gpu
DT = {"f32": (torch.float32, None), "f16": (torch.float16, np.float16)}


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_vpred_step_matches_oracle_and_reference_vectors(dt, golden_dir):
    from tweediemix_amd import ops
    z = np.load(os.path.join(golden_dir, "video_step.npz"))
    acp = z["alphas_cumprod"]
    tdt, lowp = DT[dt]
    for case in range(4):
        k = f"step.{dt}.{case}"
        t, skip, g = z[k + ".meta"]
        at, atn = O.video_alpha(acp, acp[0], int(t)), O.video_alpha(acp, acp[0], int(t) - int(skip))
        x, v = torch.from_numpy(z[k + ".x"]).to(tdt).cuda(), torch.from_numpy(z[k + ".v"]).to(tdt).cuda()
        got = ops.vpred_step(x, v, g, at, atn).float().cpu().numpy()
        want = O.video_vpred_step(z[k + ".x"], z[k + ".v"], g, at, atn, lowp)
        if lowp is None:
            np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
            np.testing.assert_allclose(got, z[k + ".out"], rtol=0, atol=2e-6)
        else:
            assert np.array_equal(got, want), np.abs(got - want).max()
            d = np.abs(got - z[k + ".out"])      # fixture: torch-CPU rounded the coefficients to fp16 first (cancellation amplifies it)
            assert d.max() <= 0.1 and (d > 2e-2 + 1e-2 * np.abs(z[k + ".out"])).mean() < 5e-3, (d.max(), k)


@pytest.mark.parametrize("dt", ["f32", "f16"])
def test_frame_inject_matches_oracle_and_reference_vectors(dt, golden_dir):
    from tweediemix_amd import ops
    z = np.load(os.path.join(golden_dir, "video_step.npz"))
    tdt, lowp = DT[dt]
    for case in range(5):
        k = f"inject.{dt}.{case}"
        hard, soft, interp, _t, active = z[k + ".meta"]
        if not active:
            continue
        x = torch.from_numpy(z[k + ".x"]).to(tdt).cuda()
        got = ops.frame_inject(x, 2, 16, None if hard else float(interp)).float().cpu().numpy()
        want = O.inject_first_frame(z[k + ".x"], 2, 16, None if hard else float(interp), lowp)
        assert np.array_equal(got, want) if lowp is not None or hard else np.allclose(got, want, rtol=0, atol=1e-6)
        np.testing.assert_allclose(got, z[k + ".out"], rtol=0, atol=1e-6 if lowp is None else 0)


def test_vpred_step_full_size_bf16_and_errors():
    """the I2VGen-XL latent [1,4,16,56,96] (768x448, 16 frames) in bf16 against the fp32 oracle, and argument checks."""
    from tweediemix_amd import ops, lib as L
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 4, 16, 56, 96, generator=g)
    v = torch.randn(2, 4, 16, 56, 96, generator=g)
    got = ops.vpred_step(x.cuda().bfloat16(), v.cuda().bfloat16(), 9.0, 0.37, 0.41).float().cpu()
    want = torch.from_numpy(O.video_vpred_step(x.bfloat16().float().numpy(), v.bfloat16().float().numpy(), 9.0, 0.37, 0.41))
    assert float((got - want).abs().max()) <= 2 ** -7 * float(want.abs().max())
    lib = L.load()
    assert lib.tmix_vpred_step(None, None, None, 0, 16, 1.0, 1.0, 0.0, 1.0, 0.0, None) != 0
    assert lib.tmix_frame_inject(x.cuda().data_ptr(), 0, 1, 1, 16, 1, 0.0, 0.0, None) != 0       # frames < 2


def test_video_sample_loop_with_injection_matches_oracle_loop(golden_dir):
    """the whole video loop (schedule, injection windows, fused update) with a stand-in v-prediction network whose
    resnet-like stage goes through FeatureInjector, against the same loop written with the oracle functions."""
    from tweediemix_amd import video as V
    acp = np.load(os.path.join(golden_dir, "video_step.npz"))["alphas_cumprod"]
    n, gs, ratio, interp = 10, 9.0, 0.3, 0.7
    sch = V.VideoSchedule(acp, n)
    assert sch.skip == 100 and list(sch.timesteps[:3]) == [901, 801, 701] and sch.injection_schedule(ratio) == {901, 801, 701}
    inj = V.FeatureInjector(sch.injection_schedule(ratio), interp)
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(1, 4, 16, 6, 5, generator=g)
    w = torch.randn(4, 4, generator=g) * 0.5

    def feats(xin, t):           # [2,C,F,H,W] -> per-frame features [(2*16), C, H, W], a 1x1 "resnet"
        f = xin.permute(0, 2, 1, 3, 4).reshape(32, 4, 6, 5)
        return torch.einsum("oc,nchw->nohw", w.to(f), f) + 0.001 * t

    def unet(xin, t):
        f = feats(xin.float(), t).contiguous()
        f = inj.apply("mid_block.resnets.0", f)
        f = inj.apply("up_blocks.1.resnets.0", torch.tanh(f).contiguous())
        return f.reshape(2, 16, 4, 6, 5).permute(0, 2, 1, 3, 4).contiguous()

    got = V.sample_loop(unet, x0.cuda(), sch, gs, inj).cpu().numpy()
    x = x0.numpy()
    for t in sch.timesteps:
        f = feats(torch.from_numpy(np.concatenate([x, x])), int(t)).numpy()
        act = V.injection_active(int(t), inj.schedule)
        if act:
            f = O.inject_first_frame(f, 2, 16, None)
        f = np.tanh(f)
        if act:
            f = O.inject_first_frame(f, 2, 16, interp)
        v = f.reshape(2, 16, 4, 6, 5).transpose(0, 2, 1, 3, 4)
        x = O.video_vpred_step(x, v, gs, sch.alpha(int(t)), sch.alpha(int(t) - sch.skip))
    np.testing.assert_allclose(got, x, rtol=1e-4, atol=1e-4)
