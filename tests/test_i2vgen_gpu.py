"""The I2VGen-XL UNet plan (tweediemix_amd/i2vgen.py) against oracle/i2vgen_oracle.py (fp32 torch restatement; parity
UNPINNED, see its header) on a tiny configuration, plus the inventory check of the full-size model."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / b.float().cpu().norm())


def _setup(cfg_o, cfg_p, B, Fr, H, W, Lk):
    from oracle import i2vgen_oracle as IO
    sd = IO.synthetic_state_dict(cfg_o)
    sd = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 else v) for k, v in sd.items()}
    g = torch.Generator().manual_seed(0)
    il = torch.randn(B, 4, Fr, H, W, generator=g)
    emb = torch.randn(B, cfg_o.cross_dim, generator=g)
    ehs = torch.randn(B, Lk, cfg_o.cross_dim, generator=g)
    fps = torch.tensor([8.0] * B)
    sample = torch.randn(B, 4, Fr, H, W, generator=g)
    return sd, il, emb, ehs, fps, sample


def test_conditioning_and_plan_match_oracle_tiny():
    from oracle import i2vgen_oracle as IO
    from tweediemix_amd import i2vgen as I
    B, Fr, H, W, Lk = 2, 16, 16, 8, 13
    sd, il, emb, ehs, fps, sample = _setup(IO.TINY, I.TINY, B, Fr, H, W, Lk)
    Wt = I.I2VWeights(I.TINY, sd)
    fe, ctx, ilf = I.conditioning(Wt, fps, il, emb, ehs)
    fe_o, ctx_o, ilf_o = IO.conditioning(sd, IO.TINY, fps, il, emb, ehs)
    assert rel(fe, fe_o) < 1e-4 and rel(ctx, ctx_o) < 1e-4 and rel(ilf, ilf_o) < 1e-4
    assert ctx.shape == (B, Lk + (IO.TINY.ctx_pool // 4) ** 2 + 4, IO.TINY.cross_dim)
    plan = I.I2VPlan(Wt, B, Fr, H, W, fe, ctx, ilf, autotune=False)
    out = plan(sample, 981).clone()
    want = IO.forward(sd, IO.TINY, sample, 981, fe_o, ctx_o, ilf_o)
    assert out.shape == want.shape == (B, 4, Fr, H, W)
    assert rel(out, want) < 2e-2, rel(out, want)
    out2 = plan(sample, 981)
    assert torch.equal(out, out2)                                  # deterministic replay


def test_full_size_inventory():
    """the restated I2VGenXLUNet has 1,420,469,224 parameters (2.84 GB in fp16, the size of the published fp16 checkpoint)
    and every key maps into the kernel-layout container."""
    from oracle import i2vgen_oracle as IO
    shapes = IO.param_shapes(IO.FULL)
    assert sum(torch.Size(s).numel() for s in shapes.values()) == 1_420_469_224


def test_full_architecture_small_frames_match_oracle():
    """the real channel / head / layer configuration (1.42 B parameters) on 2 clips x 16 frames of 16x16 latents."""
    from oracle import i2vgen_oracle as IO
    from tweediemix_amd import i2vgen as I
    B, Fr, H, W, Lk = 2, 16, 16, 16, 77
    sd, il, emb, ehs, fps, sample = _setup(IO.FULL, I.FULL, B, Fr, H, W, Lk)
    Wt = I.I2VWeights(I.FULL, sd)
    fe, ctx, ilf = I.conditioning(Wt, fps, il, emb, ehs)
    assert ctx.shape == (B, 77 + 64 + 4, 1024)
    plan = I.I2VPlan(Wt, B, Fr, H, W, fe, ctx, ilf, autotune=False)
    out = plan(sample, 501).clone()
    fe_o, ctx_o, ilf_o = IO.conditioning(sd, IO.FULL, fps, il, emb, ehs)
    want = IO.forward(sd, IO.FULL, sample, 501, fe_o, ctx_o, ilf_o)
    r = rel(out, want)
    print("I2VGen-XL full architecture rel-L2 vs fp32 restatement:", r)
    assert r < 2e-2, r


def test_full_architecture_at_768x448_16_frames_two_chains():
    """BASELINE config 5 at the size its number is quoted on: 2 clips (the CFG pair) x 16 frames of 56 x 96 latents (768 x 448),
    the real 1.42 B-parameter configuration, the two clips as two launch chains (I2VPlanGroup) replayed as a hipGraph --
    against the fp32 restatement evaluated on the same GPU."""
    from oracle import i2vgen_oracle as IO
    from tweediemix_amd import i2vgen as I
    B, Fr, H, W, Lk = 2, 16, 56, 96, 77
    sd, il, emb, ehs, fps, sample = _setup(IO.FULL, I.FULL, B, Fr, H, W, Lk)
    Wt = I.I2VWeights(I.FULL, sd)
    fe, ctx, ilf = I.conditioning(Wt, fps, il, emb, ehs)
    grp = I.I2VPlanGroup(Wt, B, Fr, H, W, fe, ctx, ilf)
    out = grp(sample, 501).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        grp.run()
    grp.eps.zero_()
    g.replay()
    torch.cuda.synchronize()
    out2 = grp.eps.view(B, Fr, 4, H, W).permute(0, 2, 1, 3, 4).clone()
    dev = lambda t: t.cuda()
    sd_g = {k: v.cuda() for k, v in sd.items()}
    fe_o, ctx_o, ilf_o = IO.conditioning(sd_g, IO.FULL, dev(fps), dev(il), dev(emb), dev(ehs))
    want = IO.forward(sd_g, IO.FULL, dev(sample), 501, fe_o, ctx_o, ilf_o)
    torch.cuda.synchronize()
    r, r2 = rel(out, want), rel(out2, want)
    print("I2VGen-XL 768x448x16 two chains: rel-L2 eager", r, "graph replay", r2)
    assert r < 2e-2 and r2 < 2e-2, (r, r2)


def test_video_loop_on_the_plan():
    """tweediemix_amd.video.sample_loop driving the plan: CFG batch of 2 clips, v-prediction update, injection hooks off."""
    from oracle import i2vgen_oracle as IO
    from tweediemix_amd import i2vgen as I, video as V
    import numpy as np
    B, Fr, H, W = 2, 16, 16, 8
    sd, il, emb, ehs, fps, sample = _setup(IO.TINY, I.TINY, B, Fr, H, W, 13)
    Wt = I.I2VWeights(I.TINY, sd)
    fe, ctx, ilf = I.conditioning(Wt, fps, il, emb, ehs)
    plan = I.I2VPlan(Wt, B, Fr, H, W, fe, ctx, ilf, autotune=False)
    acp = np.cos((np.arange(1000) / 1000 + 0.008) / 1.008 * np.pi / 2) ** 2
    sch = V.VideoSchedule(acp.astype(np.float32), 5)
    x = V.sample_loop(lambda xin, t: plan(xin, t).contiguous(), sample[:1].cuda(), sch, 9.0)
    assert x.shape == (1, 4, Fr, H, W) and torch.isfinite(x).all()


def test_feature_injection_inside_the_plan():
    """with `inject` raised the three hooked resnet outputs are overwritten like the reference's patched forward: the result
    equals the oracle forward with the same edit applied at mid_block.resnets[0,1] (hard) and up_blocks[1].resnets[0] (interp)."""
    from oracle import i2vgen_oracle as IO, tweedie_oracle as O
    from tweediemix_amd import i2vgen as I
    B, Fr, H, W = 2, 16, 16, 8
    sd, il, emb, ehs, fps, sample = _setup(IO.TINY, I.TINY, B, Fr, H, W, 13)
    Wt = I.I2VWeights(I.TINY, sd)
    fe, ctx, ilf = I.conditioning(Wt, fps, il, emb, ehs)
    plan = I.I2VPlan(Wt, B, Fr, H, W, fe, ctx, ilf, autotune=False, interp=0.7)
    base = plan(sample, 981).clone()
    plan.inject = True
    inj = plan(sample, 981).clone()
    assert rel(inj, base) > 1e-2                                     # the edit is visible
    orig = IO._resnet
    def patched(x, temb, p, n, groups):
        y = orig(x, temb, p, n, groups)
        if n in ("mid_block.resnets.0", "mid_block.resnets.1", "up_blocks.1.resnets.0"):
            y = torch.from_numpy(O.inject_first_frame(y.numpy(), 2, 16, None if n.startswith("mid") else 0.7))
        return y
    IO._resnet = patched
    try:
        fe_o, ctx_o, ilf_o = IO.conditioning(sd, IO.TINY, fps, il, emb, ehs)
        want = IO.forward(sd, IO.TINY, sample, 981, fe_o, ctx_o, ilf_o)
    finally:
        IO._resnet = orig
    assert rel(inj, want) < 2e-2, rel(inj, want)


def test_run_video_drop_in(tmp_path, monkeypatch):
    """run_video.py end to end on the tiny network: schedule, injection window (first step), CFG / v-prediction update, latent file."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("run_video_cli", os.path.join(root, "run_video.py"))
    rv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rv)
    monkeypatch.chdir(tmp_path)
    lat = rv.main(["--synthetic", "--tiny", "--height", "128", "--width", "64", "--num_inference_steps", "10", "--seed", "3",
                   "--injection_timestep", "0.2"])
    assert lat.shape == (1, 4, 16, 16, 8) and torch.isfinite(lat).all()
    assert (tmp_path / "output_i2v_seed_3.latent.pt").exists()
    lat2 = rv.main(["--synthetic", "--tiny", "--height", "128", "--width", "64", "--num_inference_steps", "10", "--seed", "3",
                    "--injection_timestep", "0.2", "--no_graphs"])
    assert torch.equal(lat.cpu(), lat2.cpu())                        # graph replay == eager


def test_run_video_from_image_and_prompt(tmp_path, monkeypatch, golden_dir):
    """run_video.py with nothing precomputed, on a synthetic diffusers-layout I2VGen-XL folder: tokenizer + text tower, CLIP image
    tower + preprocessing, VAE encoder + frame-position planes, UNet loop, VAE decode to a GIF."""
    import importlib.util, json, os, shutil
    import numpy as np
    from PIL import Image
    from safetensors.torch import save_file
    from tweediemix_amd import i2vgen as I, vae as V, weights as Wt
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ck = tmp_path / "i2v"
    cfg = I.TINY                                           # cross_dim 128
    (ck / "unet").mkdir(parents=True)
    save_file({k: v.contiguous() for k, v in Wt.synthetic_i2vgen_state_dict(cfg).items()}, str(ck / "unet" / "diffusion_pytorch_model.safetensors"))
    z = np.load(os.path.join(golden_dir, "clip_text.npz"))       # text tower: the tiny CLIP of the golden set (d = 128)
    sd = {k[len("l") + 4:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("l.sd.")}
    key = [k for k in sd if k.endswith("token_embedding.weight")][0]
    g = torch.Generator().manual_seed(1)
    sd[key] = torch.cat([sd[key], torch.randn(620 - 64, 128, generator=g) * 0.05])
    (ck / "text_encoder").mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(ck / "text_encoder" / "model.safetensors"))
    json.dump({"hidden_act": "quick_gelu", "num_attention_heads": 2, "eos_token_id": 2}, open(ck / "text_encoder" / "config.json", "w"))
    shutil.copytree(os.path.join(golden_dir, "clip_tok"), ck / "tokenizer")
    zv = np.load(os.path.join(golden_dir, "clip_vision.npz"))    # image tower: the golden tiny ViT (56x56, patch 14, proj 64 -> need 128)
    vsd = {k[3:]: torch.from_numpy(zv[k].astype(np.float32)) for k in zv.files if k.startswith("sd.")}
    vsd["visual_projection.weight"] = torch.randn(128, 320, generator=g) * 320 ** -0.5
    (ck / "image_encoder").mkdir()
    save_file({k: v.contiguous() for k, v in vsd.items()}, str(ck / "image_encoder" / "model.safetensors"))
    json.dump({"image_size": 56, "patch_size": 14, "num_attention_heads": 4, "hidden_act": "gelu"}, open(ck / "image_encoder" / "config.json", "w"))
    (ck / "vae").mkdir()
    vae_sd = V.synthetic_state_dict(V.TINY, nontrivial=True)
    vae_sd.update(V.synthetic_state_dict(V.TINY, seed=9, nontrivial=True, encoder=True))
    save_file({k: v.contiguous() for k, v in vae_sd.items()}, str(ck / "vae" / "diffusion_pytorch_model.safetensors"))
    json.dump({"block_out_channels": list(V.TINY["block_out_channels"]), "layers_per_block": 1}, open(ck / "vae" / "config.json", "w"))
    img = tmp_path / "init.png"
    Image.fromarray(np.random.RandomState(0).randint(0, 256, (200, 300, 3), dtype=np.uint8)).save(img)
    spec = importlib.util.spec_from_file_location("run_video_cli2", os.path.join(root, "run_video.py"))
    rv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rv)
    monkeypatch.chdir(tmp_path)
    lat = rv.main(["--i2v_path", str(ck), "--vae_path", str(ck / "vae"), "--image_path", str(img), "--tiny", "--height", "64", "--width", "128",
                   "--num_inference_steps", "4", "--seed", "5", "--prompt", "a cat and a dog running", "--negative_prompt", "blurry"])
    assert lat.shape == (1, 4, 16, 8, 16) and torch.isfinite(lat).all()
    gif = Image.open(tmp_path / "output_i2v_seed_5.gif")
    assert gif.n_frames == 16 and gif.size == (128, 64)


def test_plan_group_equals_single_plan():
    """the CFG pair split over two streams gives the single plan's result bit for bit (with and without injection)."""
    from oracle import i2vgen_oracle as IO
    from tweediemix_amd import i2vgen as I
    B, Fr, H, W = 2, 16, 16, 8
    sd, il, emb, ehs, fps, sample = _setup(IO.TINY, I.TINY, B, Fr, H, W, 13)
    Wt = I.I2VWeights(I.TINY, sd)
    fe, ctx, ilf = I.conditioning(Wt, fps, il, emb, ehs)
    one = I.I2VPlan(Wt, B, Fr, H, W, fe, ctx, ilf, autotune=False)
    grp = I.I2VPlanGroup(Wt, B, Fr, H, W, fe, ctx, ilf, autotune=False)
    for inj in (False, True):
        one.inject = inj
        grp.inject = inj
        a = one(sample, 701).clone()
        b = grp(sample, 701).clone()
        torch.cuda.synchronize()
        assert torch.equal(a, b), float((a - b).abs().max())


# ------------------------------------------------------------------------------------------------------------------------------
# the once-per-video conditioning kernels (csrc/conditioning.hip), each against the torch op it stands for, fp32
@pytest.mark.parametrize("B,Ci,H,W,Co,stride,silu", [(2, 4, 16, 8, 16, 1, True), (3, 16, 13, 9, 20, 1, False), (2, 32, 32, 32, 64, 2, True),
                                                     (2, 64, 16, 16, 96, 2, False), (1, 5, 7, 11, 3, 2, True), (32, 4, 56, 96, 16, 1, True)])
def test_conv3x3_f32_matches_torch(B, Ci, H, W, Co, stride, silu):
    import torch.nn.functional as F
    from tweediemix_amd import ops
    g = torch.Generator().manual_seed(B * 100 + Ci)
    x, w, b = torch.randn(B, Ci, H, W, generator=g).cuda(), (torch.randn(Co, Ci, 3, 3, generator=g) / (3 * Ci ** 0.5)).cuda(), torch.randn(Co, generator=g).cuda()
    got = ops.conv3x3_f32(x, w, b, stride=stride, silu=silu)
    want = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
    want = F.silu(want) if silu else want
    assert got.shape == want.shape and rel(got, want.float()) < 2e-6, rel(got, want.float())
    got0 = ops.conv3x3_f32(x, w, None, stride=stride)                  # no bias
    assert rel(got0, F.conv2d(x.double(), w.double(), None, stride=stride, padding=1).float()) < 2e-6


@pytest.mark.parametrize("H,W,oh,ow", [(56, 96, 32, 32), (16, 8, 8, 8), (7, 5, 3, 4), (8, 8, 8, 8), (5, 5, 7, 9)])
def test_adaptive_avgpool_f32_matches_torch(H, W, oh, ow):
    import torch.nn.functional as F
    from tweediemix_amd import ops
    x = torch.randn(2, 6, H, W, generator=torch.Generator().manual_seed(H * W)).cuda()
    got = ops.adaptive_avgpool_f32(x, oh, ow)
    assert got.shape == (2, 6, oh, ow) and (got - F.adaptive_avg_pool2d(x, (oh, ow))).abs().max().item() < 1e-6


@pytest.mark.parametrize("M,N,K", [(2, 1280, 320), (2, 4096, 1280), (20, 37, 19), (1, 5, 1024)])
def test_linear_f32_matches_torch(M, N, K):
    import torch.nn.functional as F
    from tweediemix_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = torch.randn(M, K, generator=g).cuda(), (torch.randn(N, K, generator=g) / K ** 0.5).cuda(), torch.randn(N, generator=g).cuda()
    assert rel(ops.linear_f32(x, w, b), F.linear(x.double(), w.double(), b.double()).float()) < 2e-6
    want = F.silu(F.linear(F.silu(x.double()), w.double(), b.double())).float()
    assert rel(ops.linear_f32(x, w, b, act_in=True, act_out=True), want) < 2e-6
    assert rel(ops.linear_f32(x, w), F.linear(x.double(), w.double()).float()) < 2e-6


@pytest.mark.parametrize("clips,frames,H,W", [(2, 16, 16, 8), (1, 9, 5, 7), (2, 16, 56, 96), (3, 1, 4, 4)])
def test_i2v_temporal_encoder_matches_torch(clips, frames, H, W):
    """the fused image_latents_temporal_encoder launch against the block written out in torch (LayerNorm, two heads of 4 over the frames,
    out-projection + residual, exact-GELU feed-forward + residual; I2VGenXLTransformerTemporalEncoder), fp64 reference."""
    import torch.nn.functional as F
    from tweediemix_amd import ops
    g = torch.Generator().manual_seed(clips * 7 + frames)
    C_ = 4
    n = "e"
    shapes = {".norm1.weight": (C_,), ".norm1.bias": (C_,), ".attn1.to_q.weight": (2 * C_, C_), ".attn1.to_k.weight": (2 * C_, C_), ".attn1.to_v.weight": (2 * C_, C_),
              ".attn1.to_out.0.weight": (C_, 2 * C_), ".attn1.to_out.0.bias": (C_,), ".ff.net.0.proj.weight": (4 * C_, C_), ".ff.net.0.proj.bias": (4 * C_,),
              ".ff.net.2.weight": (C_, 4 * C_), ".ff.net.2.bias": (C_,)}
    p = {n + k: (torch.randn(*s, generator=g) * (1.0 if len(s) == 1 else 0.7)).cuda() for k, s in shapes.items()}
    x = (torch.randn(clips * frames, C_, H, W, generator=g) * 2 + 0.5).cuda()
    got = ops.i2v_temporal_encoder(x, clips, frames, p, n)
    d = {k: v.double() for k, v in p.items()}
    t = x.double().view(clips, frames, C_, H, W).permute(0, 3, 4, 1, 2).reshape(clips * H * W, frames, C_)
    h = F.layer_norm(t, (C_,), d[n + ".norm1.weight"], d[n + ".norm1.bias"], 1e-5)
    q, k, v = [F.linear(h, d[n + f".attn1.to_{c}.weight"]).view(-1, frames, 2, C_).transpose(1, 2) for c in "qkv"]
    o = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(-1, frames, 2 * C_)
    t = t + F.linear(o, d[n + ".attn1.to_out.0.weight"], d[n + ".attn1.to_out.0.bias"])
    t = t + F.linear(F.gelu(F.linear(t, d[n + ".ff.net.0.proj.weight"], d[n + ".ff.net.0.proj.bias"])), d[n + ".ff.net.2.weight"], d[n + ".ff.net.2.bias"])
    want = t.view(clips, H, W, frames, C_).permute(0, 4, 3, 1, 2).float()
    assert got.shape == want.shape == (clips, C_, frames, H, W) and rel(got, want) < 1e-5, rel(got, want)


def test_conditioning_runs_on_the_library_only(monkeypatch):
    """conditioning() must not reach MIOpen / SDPA: the torch functional ops it used to call raise while it runs, and the result still
    matches the oracle."""
    import torch.nn.functional as F
    from oracle import i2vgen_oracle as IO
    from tweediemix_amd import i2vgen as I
    B, Fr, H, W, Lk = 2, 16, 16, 8, 13
    sd, il, emb, ehs, fps, sample = _setup(IO.TINY, I.TINY, B, Fr, H, W, Lk)
    Wt = I.I2VWeights(I.TINY, sd)
    fe_o, ctx_o, ilf_o = IO.conditioning(sd, IO.TINY, fps, il, emb, ehs)

    def boom(*a, **k):
        raise AssertionError("conditioning() called a torch functional op")
    for name in ("conv2d", "linear", "scaled_dot_product_attention", "layer_norm", "adaptive_avg_pool2d", "gelu", "silu"):
        monkeypatch.setattr(F, name, boom)
    fe, ctx, ilf = I.conditioning(Wt, fps, il, emb, ehs)
    assert rel(fe, fe_o) < 1e-4 and rel(ctx, ctx_o) < 1e-4 and rel(ilf, ilf_o) < 1e-4


def test_groupnorm_statistics_from_producers_match_the_statistics_kernel(monkeypatch):
    """the video plan takes the GroupNorm statistics of the per-frame norms (HW % 32 == 0) and of the clip-wide norms (TemporalConvLayer, TransformerTemporalModel)
    from the producers' column partials; TMIX_GN_STATS_KERNEL=1 plans the three-launch form everywhere.  Same network, same weights; the injection sites keep the statistics kernel."""
    from oracle import i2vgen_oracle as IO
    from tweediemix_amd import i2vgen as I
    B, Fr, H, W, Lk = 2, 16, 16, 8, 13
    sd, il, emb, ehs, fps, sample = _setup(IO.TINY, I.TINY, B, Fr, H, W, Lk)
    Wt = I.I2VWeights(I.TINY, sd)
    fe, ctx, ilf = I.conditioning(Wt, fps, il, emb, ehs)
    fused = I.I2VPlan(Wt, B, Fr, H, W, fe, ctx, ilf, autotune=False)
    monkeypatch.setenv("TMIX_GN_STATS_KERNEL", "1")
    plain = I.I2VPlan(Wt, B, Fr, H, W, fe, ctx, ilf, autotune=False)
    names = lambda p: [getattr(fn, "__name__", "") for fn, _a in p.ops]
    nf, npl = names(fused), names(plain)
    assert npl.count("tmix_groupnorm_nhwc_pre") == 0 and nf.count("tmix_groupnorm_nhwc_pre") > nf.count("tmix_groupnorm_nhwc") > 0
    assert nf.count("tmix_groupnorm_nhwc_pre") + nf.count("tmix_groupnorm_nhwc") == npl.count("tmix_groupnorm_nhwc")
    # two evaluation orders of this bf16 network sit 1.4e-2 - 1.6e-2 apart whatever the difference is (the same plan under two tilings: 1.6e-2), each 1.4e-2 from
    # the fp32 oracle: the two GroupNorm forms must be no further apart than that, with and without the injection
    for inject in (False, True):
        fused.inject = plain.inject = inject
        a, b = fused(sample, 981).clone(), plain(sample, 981).clone()
        assert rel(a, b) < 2.5e-2, (inject, rel(a, b))
    want = IO.forward(sd, IO.TINY, sample, 981, *IO.conditioning(sd, IO.TINY, fps, il, emb, ehs))
    fused.inject = plain.inject = False
    assert rel(fused(sample, 981), want) < 2e-2 and rel(plain(sample, 981), want) < 2e-2
