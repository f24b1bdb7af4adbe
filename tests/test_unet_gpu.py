"""GPU: the HIP UNet plan against the fp32 torch oracle of the same architecture and weights.

Tolerance (stated): the plan keeps activations in bf16 between kernels (8 mantissa bits; the reference's
fp16 autocast has 11) with fp32 accumulation; against the fp32 oracle on identical bf16-rounded weights
the noise prediction must agree to a relative L2 error <= 2e-2 and max-abs error <= 5e-2 * max|ref|.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    return ((a - b).norm() / b.norm()).item()


def make(kind, B, h, w, routed, seed=0, lora_mode="merged", hostile=False):
    from oracle import unet_oracle as UO
    from tweediemix_amd import unet as U, weights as Wt
    cfg = U.TINY
    sd = Wt.synthetic_state_dict(cfg, seed=1234, nontrivial=True, hostile=hostile)
    K = 3
    con = Wt.synthetic_concepts(cfg, kind, K) if kind != "none" else None
    g = torch.Generator().manual_seed(seed)
    ehs = torch.randn(B, 77, cfg.cross_dim, generator=g).to(torch.bfloat16).float()
    pooled = torch.randn(B, cfg.pooled_dim, generator=g)
    time_ids = torch.tensor([[h * 8, w * 8, 0, 0, h * 8, w * 8]] * B, dtype=torch.float32)
    x = torch.randn(1, 4, h, w, generator=g).repeat(B, 1, 1, 1)
    # oracle
    oc = UO.Concepts()
    if kind == "custom":
        oc = UO.Concepts("custom", kv={tb: [(c[f"{tb}.attn2.to_k.weight"], c[f"{tb}.attn2.to_v.weight"]) for c in con]
                                       for tb in UO.attention_prefixes(UO.TINY)})
    elif kind == "lora":
        lo = {}
        for tb in UO.attention_prefixes(UO.TINY):
            for a in ("attn1", "attn2"):
                lo[f"{tb}.{a}"] = [{nm: (c[f"{tb}.{a}.processor.to_{nm}_lora.down.weight"], c[f"{tb}.{a}.processor.to_{nm}_lora.up.weight"])
                                    for nm in ("q", "k", "v", "out")} for c in con]
        oc = UO.Concepts("lora", lora=lo)
    orc = UO.UNetOracle(UO.TINY, sd, oc)
    W = U.UNetWeights(cfg, sd, "cuda", (kind, con) if con else None, lora_mode=lora_mode)
    wsel = list(range(B)) if (routed and B == 4) else [0] * B
    kv = U.KVCache(W, ehs, wsel)
    plan = U.UNetPlan(W, B, h, w, kv, pooled, time_ids, routed=routed)
    return orc, plan, x, ehs, pooled, time_ids


@pytest.mark.parametrize("kind,B,routed", [("none", 2, False), ("custom", 4, True), ("custom", 4, False),
                                           ("lora", 4, True), ("lora", 4, False), ("lora", 2, True)])
@pytest.mark.parametrize("hw", [(16, 16), (8, 24)])
def test_unet_plan_matches_oracle(kind, B, routed, hw):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    h, w = hw
    orc, plan, x, ehs, pooled, time_ids = make(kind, B, h, w, routed)
    t = 781
    ref = orc.forward(x, t, ehs, pooled, time_ids, routed=routed)
    eps = plan(x.cuda(), t).float().cpu()
    torch.cuda.synchronize()
    assert torch.isfinite(eps).all()
    r = rel_l2(eps, ref)
    m = (eps - ref).abs().max().item() / ref.abs().max().item()
    print(f"{kind} B={B} routed={routed} {hw}: rel_l2={r:.4g} max_rel={m:.4g} flops={plan.flops:.3g} launches={len(plan.ops)}")
    assert r <= 2e-2 and m <= 5e-2, (r, m)
    # replay is deterministic and allocation-free
    eps2 = plan(x.cuda(), t).float().cpu()
    assert torch.equal(eps, eps2)


@pytest.mark.parametrize("kind,hw", [("lora", (32, 32)), ("custom", (16, 16))])
def test_groupnorm_statistics_from_the_producers_match_the_statistics_kernel(kind, hw, monkeypatch):
    """the default plan takes every GroupNorm's statistics from the conv / proj_out launch that wrote the tensor (col_stats_out ->
    tmix_groupnorm_nhwc_pre; concatenations carry their two sources' partials); TMIX_GN_STATS_KERNEL=1 is the three-launch form.
    Same network, same weights: the two differ only in the summation order of the statistics."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    h, w = hw
    orc, plan, x, ehs, pooled, time_ids = make(kind, 4, h, w, True)
    names = [fn.__name__ for fn, _a in plan.ops]
    n_pre, n_old = names.count("tmix_groupnorm_nhwc_pre"), names.count("tmix_groupnorm_nhwc")
    assert n_pre > 0 and (h % 32 or n_old <= 2), (n_pre, n_old)        # at 32 x 32 only conv_in's consumers keep the statistics kernel (level 2 is 8 x 8 = 64 pixels: fused too)
    eps = plan(x.cuda(), 500).float().cpu()
    monkeypatch.setenv("TMIX_GN_STATS_KERNEL", "1")
    _orc, plan0, *_ = make(kind, 4, h, w, True)
    assert "tmix_groupnorm_nhwc_pre" not in [fn.__name__ for fn, _a in plan0.ops]
    eps0 = plan0(x.cuda(), 500).float().cpu()
    ref = orc.forward(x, 500, ehs, pooled, time_ids, routed=True)
    print(f"{kind} {hw}: {n_pre} norms from producer partials, {n_old} with their own statistics pass; fused vs three-launch rel_l2={rel_l2(eps, eps0):.3g}, "
          f"vs oracle {rel_l2(eps, ref):.3g} / {rel_l2(eps0, ref):.3g}")
    assert rel_l2(eps, ref) <= 2e-2 and rel_l2(eps, eps0) <= 1e-2


def test_weight_hints_are_results_neutral_and_obey_the_cap(monkeypatch):
    """tmix_gemm_prefetch_next names (part of) the next launch's weights (UNetPlan._hint_weights, unet.hint_policy): with whole-tensor hints, with a cap that bites on
    this small network's tensors, and with no hints at all the call returns the same bits; every capped hint names at most the cap."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    def hints(plan):
        return [a for fn, a in plan.ops if fn.__name__ == "tmix_gemm_prefetch_next"]
    monkeypatch.setenv("TMIX_PF_CAP_MB", "0")
    orc, plan, x, ehs, pooled, time_ids = make("lora", 4, 16, 16, True)
    whole = [a[1] for a in hints(plan) if a[0]]
    assert whole and max(whole) > 8192
    eps_whole = plan(x.cuda(), 500).clone()
    cap = 4096
    monkeypatch.setenv("TMIX_PF_CAP_MB", str(cap / (1 << 20))); monkeypatch.setenv("TMIX_PF_CAP_OVER_MB", str(8192 / (1 << 20)))
    _o, plan_c, *_ = make("lora", 4, 16, 16, True)
    capped = [a[1] for a in hints(plan_c) if a[0]]
    assert len(capped) == len(whole) and all(c == (cap if wb > 8192 else wb) for c, wb in zip(capped, whole)) and min(capped) < max(whole)
    eps_capped = plan_c(x.cuda(), 500).clone()
    monkeypatch.delenv("TMIX_PF_CAP_MB"); monkeypatch.delenv("TMIX_PF_CAP_OVER_MB"); monkeypatch.setenv("TMIX_NO_PREFETCH", "1")
    _o, plan_n, *_ = make("lora", 4, 16, 16, True)
    assert not hints(plan_n)
    eps_none = plan_n(x.cuda(), 500).clone()
    assert torch.equal(eps_whole, eps_capped) and torch.equal(eps_whole, eps_none)
    ref = orc.forward(x, 500, ehs, pooled, time_ids, routed=True)
    assert rel_l2(eps_whole.float().cpu(), ref) <= 2e-2


@pytest.mark.parametrize("kind,hw", [("lora", (32, 32)), ("none", (8, 24))])
def test_shortcut_in_the_conv_launch_matches_the_separate_gemm_and_concat(kind, hw, monkeypatch):
    """the default plan runs conv_shortcut inside conv2's launch (shortcut taps) and never writes the up-blocks' concatenations;
    TMIX_SHORTCUT_GEMM=1 is the form with a shortcut GEMM + residual and concat launches.  Same network, same weights."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    h, w = hw
    B = 4 if kind == "lora" else 2
    orc, plan, x, ehs, pooled, time_ids = make(kind, B, h, w, kind == "lora")
    names = [fn.__name__ for fn, _a in plan.ops]
    assert "tmix_concat_channels" not in names
    eps = plan(x.cuda(), 500).float().cpu()
    monkeypatch.setenv("TMIX_SHORTCUT_GEMM", "1")
    _orc, plan0, *_ = make(kind, B, h, w, kind == "lora")
    names0 = [fn.__name__ for fn, _a in plan0.ops]
    assert names0.count("tmix_concat_channels") == 9 and len(names0) > len(names)
    eps0 = plan0(x.cuda(), 500).float().cpu()
    ref = orc.forward(x, 500, ehs, pooled, time_ids, routed=(kind == "lora"))
    print(f"{kind} {hw}: {len(names)} vs {len(names0)} ops; fused vs separate rel_l2={rel_l2(eps, eps0):.3g}, vs oracle {rel_l2(eps, ref):.3g} / {rel_l2(eps0, ref):.3g}")
    assert rel_l2(eps, ref) <= 2e-2 and rel_l2(eps, eps0) <= 1e-2


@pytest.mark.parametrize("hw", [(16, 16), (8, 24)])
def test_unet_plan_with_lora_in_low_rank_form_matches_oracle_and_the_merged_plan(hw):
    """UNetWeights(lora_mode="lowrank"): up(down(x)) as the routed projections' last K-tile (tmix_lora_down fills the pad columns,
    shared weights [W | U | 0]; utils_lora.py:68,76-77,118) -- same bound against the fp32 oracle as the merged form, no per-concept
    weight copies, and the two forms agree with each other far inside that bound."""
    h, w = hw
    orc, plan, x, ehs, pooled, time_ids = make("lora", 4, h, w, True, lora_mode="lowrank")
    assert plan.lowrank and not any(k.endswith("_rows") and "kv_rows" not in k for k in plan.W.t)
    t = 781
    ref = orc.forward(x, t, ehs, pooled, time_ids, routed=True)
    eps = plan(x.cuda(), t).float().cpu()
    r = rel_l2(eps, ref)
    m = (eps - ref).abs().max().item() / ref.abs().max().item()
    _orc, plan_m, *_ = make("lora", 4, h, w, True)
    eps_m = plan_m(x.cuda(), t).float().cpu()
    print(f"low-rank LoRA {hw}: rel_l2={r:.4g} max_rel={m:.4g}; vs merged plan rel_l2={rel_l2(eps, eps_m):.4g}; launches {len(plan.ops)} vs {len(plan_m.ops)}")
    assert r <= 2e-2 and m <= 5e-2, (r, m)
    assert rel_l2(eps, eps_m) <= 1e-2
    # an unrouted call of the same weights (outside the window): no deltas at all == the base network
    orc0, plan0, *_ = make("lora", 4, h, w, False, lora_mode="lowrank")
    assert not plan0.lowrank
    ref0 = orc0.forward(x, t, ehs, pooled, time_ids, routed=False)
    assert rel_l2(plan0(x.cuda(), t).float().cpu(), ref0) <= 2e-2


@pytest.mark.parametrize("scale", [1.0, 0.01])
def test_lora_low_rank_form_resolves_deltas_below_the_ulp_of_w(scale):
    """the same 1280-wide projection as test_lora_delta_merged_into_bf16_weights through the low-rank path (tmix_lora_down + one GEMM over
    K + 64 columns): the delta's own contribution -- low-rank output minus base output -- is exact to bf16 operand rounding (~1 %) at the
    checkpoints' delta size AND at deltas 100x smaller, where the merged form only dithers it (~0.9 relative error)."""
    from tweediemix_amd import ops
    g = torch.Generator().manual_seed(21)
    M, N, K = 1024, 1280, 1280
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W32 = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    down = (torch.randn(4, K, generator=g) * 0.25).cuda()
    up = (torch.randn(N, 4, generator=g) * 0.02 * scale).cuda()
    y_delta = (x.float() @ down.T) @ up.T
    def f32_gemm(a_, w_):                                         # fp32 C (TMIX_EPI_F32OUT): the OUTPUT rounding to bf16 would bury a small delta in either form
        o = torch.empty(a_.shape[0], w_.shape[0], dtype=torch.float32, device="cuda")
        ops.gemm(a_, w_, out_f32=o)
        return o
    y_base = f32_gemm(x, W32.to(torch.bfloat16))
    y_merged = f32_gemm(x, (W32 + up @ down).to(torch.bfloat16))
    a = torch.zeros(M, K + 64, dtype=torch.bfloat16).cuda()
    a[:, :K] = x
    D = torch.zeros(8, K).cuda()
    D[4:] = down                                                   # set 0 = base (zeros), set 1 = the concept
    U = torch.zeros(N, 64).cuda()
    U[:, 4:8] = up
    ops.lora_down(a, K, D.to(torch.bfloat16).contiguous(), 4, 2, torch.tensor([1], dtype=torch.int32).cuda(), M)
    y_lr = f32_gemm(a, torch.cat([W32, U], 1).to(torch.bfloat16).contiguous())
    e_lr = rel_l2(y_lr - y_base, y_delta)
    e_m = rel_l2(y_merged - y_base, y_delta)
    print(f"delta scale {scale}: contribution error low-rank {e_lr:.3g}, merged {e_m:.3g}  (|delta|/|W| = {(up @ down).abs().mean().item() / W32.abs().mean().item():.2g})")
    assert e_lr < 2e-2                                             # bf16 rounding of T = x down^T and of up: ~2^-9 each
    if scale < 1.0:
        assert e_m > 0.5                                           # the merged form has only dithered the delta into W


def test_kv_cache_routing_exact():
    """row b of the cached cross-attention K / V^T must come from weight set wsel[b]
    (utils_custom.py:64-83: row 0 base to_k/to_v, row 1+i concept i's) -- checked per layer against torch."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tweediemix_amd import unet as U, weights as Wt
    cfg = U.TINY
    sd = Wt.synthetic_state_dict(cfg, seed=1234, nontrivial=True)
    con = Wt.synthetic_concepts(cfg, "custom", 3)
    W = U.UNetWeights(cfg, sd, "cuda", ("custom", con))
    g = torch.Generator().manual_seed(1)
    ehs = torch.randn(4, 77, cfg.cross_dim, generator=g).to(torch.bfloat16)
    for wsel in ([0, 1, 2, 3], [0, 0, 0, 0], [0, 3, 1, 2]):
        kv = U.KVCache(W, ehs, wsel)
        for tb, Cc in U.attention_blocks(cfg):
            a2 = tb + ".attn2"
            for b, ws in enumerate(wsel):
                src = sd if ws == 0 else con[ws - 1]
                kref = ehs[b].float() @ src[a2 + ".to_k.weight"].float().T
                vref = ehs[b].float() @ src[a2 + ".to_v.weight"].float().T
                torch.testing.assert_close(kv.k[a2][b].float().cpu(), kref, rtol=2 ** -7, atol=2e-2)
                torch.testing.assert_close(kv.vt[a2][b, :, :77].float().cpu(), vref.T, rtol=2 ** -7, atol=2e-2)
                assert (kv.vt[a2][b, :, 77:] == 0).all()
    # different weight sets really are different (the check above is sensitive)
    a2 = U.attention_blocks(cfg)[0][0] + ".attn2"
    assert (sd[a2 + ".to_k.weight"] - con[0][a2 + ".to_k.weight"]).abs().max() > 0.05


def test_lora_merged_rows():
    """LoRA rows: weight set 1+i = W + up_i @ down_i (utils_lora.py:65-79,113-119; SURVEY 3.2 probe)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tweediemix_amd import unet as U, weights as Wt
    cfg = U.TINY
    sd = Wt.synthetic_state_dict(cfg, seed=1234)
    con = Wt.synthetic_concepts(cfg, "lora", 3)
    W = U.UNetWeights(cfg, sd, "cuda", ("lora", con))
    tb, Cc = U.attention_blocks(cfg)[3]
    a1 = tb + ".attn1"
    rows = W[a1 + ".qkv_rows"].float().cpu()
    assert rows.shape == (4, 3 * Cc, Cc)
    for i in range(3):
        for j, nm in enumerate(("q", "k", "v")):
            ref = sd[f"{a1}.to_{nm}.weight"] + con[i][f"{a1}.processor.to_{nm}_lora.up.weight"] @ con[i][f"{a1}.processor.to_{nm}_lora.down.weight"]
            torch.testing.assert_close(rows[i + 1, j * Cc:(j + 1) * Cc], ref, rtol=2 ** -7, atol=1e-3)
    torch.testing.assert_close(rows[0, :Cc], sd[f"{a1}.to_q.weight"], rtol=2 ** -7, atol=1e-3)


@pytest.mark.parametrize("scale", [1.0, 0.01])
def test_lora_delta_merged_into_bf16_weights(scale):
    """the routed LoRA projections run on merged weights bf16(W + up @ down) (UNetWeights.merged) where the reference adds
    up(down(x)) as a separate fp16 path (utils_lora.py:68,76-77,118; model_lora.py:41-48).  What that costs, measured on one
    1280-wide projection: (a) the OUTPUT error against the exact fp32 x W^T + x down^T up^T must not exceed the error a bf16 copy
    of W alone already has (rounding W + delta leaves the same e as rounding W) -- asserted at the synthetic checkpoints' delta size
    and at deltas 100x smaller (|delta| ~ 1e-4, the ulp of |W| ~ 0.03); (b) the delta's own contribution, recovered as merged
    minus base output: exact to ~1 % at the checkpoints' size, but only dithered at the 100x smaller size (printed: ~0.9 relative
    error, i.e. below the bf16-weight noise floor that ANY bf16-W form of this GEMM has, the rank-4 epilogue form included; that
    form would resolve the delta itself exactly).  DESIGN.md section 4.2 prices the epilogue form."""
    from tweediemix_amd import ops
    g = torch.Generator().manual_seed(21)
    M, N, K = 1024, 1280, 1280
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    W32 = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
    down = (torch.randn(4, K, generator=g) * 0.25).cuda()
    up = (torch.randn(N, 4, generator=g) * 0.02 * scale).cuda()            # scale 1 = weights.synthetic_concepts
    delta = up @ down
    y_delta = (x.float() @ down.T) @ up.T
    y_exact = x.float() @ W32.T + y_delta
    y_base = ops.gemm(x, W32.to(torch.bfloat16)).float()
    y_merged = ops.gemm(x, (W32 + delta).to(torch.bfloat16)).float()
    torch.cuda.synchronize()
    rec = float(((y_merged - y_base) - y_delta).norm() / y_delta.norm())
    err_merged = float((y_merged - y_exact).norm() / y_exact.norm())
    err_base = float((y_base - x.float() @ W32.T).norm() / y_exact.norm())
    print(f"|delta|/ulp(W) = {float(delta.abs().mean() / (W32.abs().mean() * 2.0 ** -8)):.2f}: delta contribution recovered to rel. error "
          f"{rec:.3f}; output error merged {err_merged:.2e} vs base-weights-only {err_base:.2e}")
    assert err_merged < 1.25 * err_base + 1e-4, (err_merged, err_base)
    if scale == 1.0:
        assert rec < 0.05, rec


@pytest.mark.parametrize("force_tile", [1, 0])
def test_plan_group_row_split_matches_single_plan(monkeypatch, force_tile):
    """PlanGroup (rows split over HIP streams) must give the rows a single plan gives: bit for bit when both use the same
    tiling (every kernel is batch-invariant), and to bf16 rounding when each picks its own (a chain that shares the chip
    is tuned differently from one that runs alone, and the tiling sets the order of the LayerNorm partial sums)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if force_tile:
        monkeypatch.setenv("TMIX_FORCE_TILE", str(force_tile))
    from tweediemix_amd import unet as U, weights as Wt
    cfg = U.TINY
    sd = Wt.synthetic_state_dict(cfg, seed=1234, nontrivial=True)
    con = Wt.synthetic_concepts(cfg, "lora", 3)
    W = U.UNetWeights(cfg, sd, "cuda", ("lora", con))
    g = torch.Generator().manual_seed(2)
    B, h, w = 4, 16, 16
    ehs = torch.randn(B, 77, cfg.cross_dim, generator=g)
    pooled = torch.randn(B, cfg.pooled_dim, generator=g)
    tid = torch.tensor([[128.0, 128, 0, 0, 128, 128]] * B)
    x = torch.randn(B, 4, h, w, generator=g).cuda()
    wsel = [0, 1, 2, 3]
    one = U.UNetPlan(W, B, h, w, U.KVCache(W, ehs, wsel), pooled, tid, routed=True, row_sets=wsel)
    ref = one(x, 500).clone()
    for ng in (2, 4):
        grp = U.PlanGroup(W, h, w, ehs, wsel, pooled, tid, True, ng)
        grp.latent.copy_(x)
        grp.t_dev.fill_(500.0)
        grp.run()
        torch.cuda.synchronize()
        if force_tile:
            assert torch.equal(grp.eps, ref)
        else:
            assert float((grp.eps - ref).norm() / ref.norm()) < 5e-3


def test_full_size_sdxl_unet_vs_fp32_oracle_on_gpu():
    """the REAL SDXL-base shapes (2.57 B parameters, 70 transformer blocks, C up to 1280, 20 heads) at 512x512,
    B = K+1 = 4 with concept-routed K/V, against the fp32 torch oracle evaluated on the same GPU.
    Exercises every autotuned tiling, the 23040-deep convolutions and the S=4096/1024 attention at production size.
    Tolerance: rel L2 <= 2e-2, max-abs <= 5e-2 * max|ref| (bf16 activations through 70 blocks vs fp32)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import unet_oracle as UO
    from tweediemix_amd import unet as U, weights as Wt
    cfg = U.SDXL
    dev = "cuda"
    sd = Wt.synthetic_state_dict(cfg, seed=1234, device=dev, dtype=torch.float32)
    con = Wt.synthetic_concepts(cfg, "custom", 3, device=dev)
    g = torch.Generator().manual_seed(3)
    B, h, w = 4, 64, 64
    ehs = torch.randn(B, 77, cfg.cross_dim, generator=g).to(torch.bfloat16).float()
    pooled = torch.randn(B, cfg.pooled_dim, generator=g)
    tid = torch.tensor([[512.0, 512, 0, 0, 512, 512]] * B)
    x = torch.randn(1, 4, h, w, generator=g).repeat(B, 1, 1, 1)
    W = U.UNetWeights(cfg, sd, dev, ("custom", con))
    plan = U.UNetPlan(W, B, h, w, U.KVCache(W, ehs, [0, 1, 2, 3]), pooled, tid)
    eps = plan(x.cuda(), 601).clone()
    oc = UO.Concepts("custom", kv={tb: [(c[f"{tb}.attn2.to_k.weight"], c[f"{tb}.attn2.to_v.weight"]) for c in con]
                                   for tb in UO.attention_prefixes(UO.SDXL)})
    ref = UO.UNetOracle(UO.SDXL, sd, oc).forward(x.cuda(), 601, ehs.cuda(), pooled.cuda(), tid.cuda(), routed=True)
    torch.cuda.synchronize()
    assert torch.isfinite(eps).all()
    r = rel_l2(eps, ref)
    m = (eps - ref).abs().max().item() / ref.abs().max().item()
    print(f"SDXL full-size 512^2 B=4: rel_l2={r:.4g} max_rel={m:.4g}")
    assert r <= 2e-2 and m <= 5e-2, (r, m)


# ------------------------------------------------------------------------------------------------ headline sizes
# (`sdxl_weights`: the session fixture of tests/conftest.py -- the real SDXL-base parameter shapes, synthetic values)


def _oracle_concepts(kind, con):
    from oracle import unet_oracle as UO
    if kind == "custom":
        return UO.Concepts("custom", kv={tb: [(c[f"{tb}.attn2.to_k.weight"], c[f"{tb}.attn2.to_v.weight"]) for c in con]
                                         for tb in UO.attention_prefixes(UO.SDXL)})
    lo = {}
    for tb in UO.attention_prefixes(UO.SDXL):
        for a in ("attn1", "attn2"):
            lo[f"{tb}.{a}"] = [{nm: (c[f"{tb}.{a}.processor.to_{nm}_lora.down.weight"], c[f"{tb}.{a}.processor.to_{nm}_lora.up.weight"])
                                for nm in ("q", "k", "v", "out")} for c in con]
    return UO.Concepts("lora", lora=lo)


def _timed_plan(sd, kind, hw, streams, fp8=False, lora_mode="merged", call_kind="fusion", n_seeds=1, bundle=None):
    """a UNet plan EXACTLY as bench.py's timed regions build it: bench.build_sampler's Tweediemix (same flags, concept routing,
    `streams` launch chains, `n_seeds` co-batched trajectories, shipped tile table) -> tw.plan(call_kind), i.e.
    sampler.Tweediemix._build_plan.  call_kind "fusion" (the headline step: uncond + K concept rows, routed), "fusion_base" (the LoRA
    window's off-by-one step at t_stop: same rows, base weights), "start" (uncond, multi, K - 1 single-concept rows; un-routed) or
    "plain" (the B = 2 CFG pair).  Returns (plan, ehs, pooled, time_ids, concept state dicts); the prompt rows are ONE seed's (co-batched seeds repeat them)."""
    from tweediemix_amd import masks as M, sampler as S, unet as U, weights as Wt
    cfg, K = U.SDXL, 3
    if bundle is not None:                               # the session's (concepts, UNetWeights, oracle) for this kind (tests/conftest.py: built once)
        con, W = bundle[0], bundle[1]
        assert lora_mode == "merged" and W.kind == kind
    else:
        con = Wt.synthetic_concepts(cfg, kind, K, device="cuda")
        W = U.UNetWeights(cfg, sd, "cuda", (kind, con), lora_mode=lora_mode)
    g = torch.Generator().manual_seed(5)
    te = (torch.randn(K + 2, 77, cfg.cross_dim, generator=g).to(torch.bfloat16).float(), torch.randn(K + 2, cfg.pooled_dim, generator=g))
    ts = (torch.randn(K, 77, cfg.cross_dim, generator=g).to(torch.bfloat16).float(), torch.randn(K, cfg.pooled_dim, generator=g))
    res = hw * 8
    conf = S.make_config(guidance_scale=0.8, n_timesteps=50, t_cond=0.2, t_stop=0.8, resampling_steps=10, jumping_steps=5,
                         resolution_h=res, resolution_w=res, seed=0)
    tw = S.Tweediemix(conf, W, te, ts, lambda x0: M.build_masks(M.random_rectangle_masks(K, res, res, seed=1), hw, hw, "cuda"),
                      concept_num=K, lora=(kind == "lora"), use_graphs=True, n_seeds=n_seeds, n_streams=streams, fp8=fp8)
    tw.init_fusion(10, 40) if kind == "lora" else tw.init_fusion(10)
    plan = tw.plan(call_kind)
    if call_kind in ("fusion", "fusion_base"):           # fusion_sampling.py:324-340
        ehs = torch.cat([te[0][0:1], te[0][2:2 + K]])
        pooled = torch.cat([te[1][0:1], te[1][2:2 + K]])
    elif call_kind == "start":                           # :342-359
        ehs = torch.cat([te[0][0:2], ts[0][1:K]])
        pooled = torch.cat([te[1][0:2], ts[1][1:K]])
    else:                                                # :360-374
        ehs, pooled = te[0][0:2], te[1][0:2]
    tid = tw.add_time_ids.repeat(ehs.shape[0], 1)
    return plan, ehs, pooled, tid, con


def _graph_replay(plan, x, t):
    plan.latent.copy_(x)
    plan.t_dev.fill_(float(t))
    plan.run()                                           # warm-up outside capture
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        plan.run()
    plan.eps.zero_()
    gr.replay()
    torch.cuda.synchronize()
    return plan.eps.clone()


@pytest.mark.parametrize("streams", [1, 2])
@pytest.mark.parametrize("kind,hw", [("custom", 128), ("lora", 128), ("lora", 64)])
def test_headline_size_timed_plan_graph_vs_fp32_oracle(sdxl_weights, sdxl_bundles, kind, hw, streams):
    """what bench.py TIMES, checked against the oracle: SDXL shapes at latent 128x128 (1024x1024: S = 16384 / 4096 / 1024 tokens,
    65,536-row convolutions), B = K+1 = 4 concept-routed rows, built by the sampler's own plan builder (see _timed_plan) --
    streams = 1: the DEFAULT one launch chain at the full batch (routed LoRA `row_sets`, un-prefixed tile-table entries);
    streams = 2: two B = 2 chains on two streams (`shared|` entries) -- captured into a hipGraph and replayed, against the fp32
    torch oracle on the same GPU.  The tilings are asserted to be the shipped table's (nothing re-tuned on this box), which is
    what bench.py asserts for its timed plan too.  Tolerance as everywhere: rel L2 <= 2e-2, max-abs <= 5e-2 * max|ref|."""
    from oracle import unet_oracle as UO
    from tweediemix_amd import unet as U
    plan, ehs, pooled, tid, con = _timed_plan(sdxl_weights, kind, hw, streams, bundle=sdxl_bundles(kind))
    assert isinstance(plan, U.PlanGroup) == (streams == 2)
    assert all(p.routed == (kind == "lora") for p in U._plans_of(plan))
    if hw == 128:                                        # the size bench.py times: every launch shape is in the shipped table
        ok, bad = U.tilings_follow_table(plan)
        assert ok, bad[:5]
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 4, hw, hw, generator=g).repeat(4, 1, 1, 1).cuda()
    eps = _graph_replay(plan, x, 601)
    ref = sdxl_bundles(kind)[2].forward(x, 601, ehs.cuda(), pooled.cuda(), tid.cuda(), routed=True)
    torch.cuda.synchronize()
    r = rel_l2(eps, ref)
    m = (eps - ref).abs().max().item() / ref.abs().max().item()
    print(f"SDXL {kind} {hw * 8}^2 B=4, {streams} chain(s), graph replay: rel_l2={r:.4g} max_rel={m:.4g} tilings={U.used_tilings(plan)}")
    assert torch.isfinite(eps).all() and r <= 2e-2 and m <= 5e-2, (r, m)


# every OTHER call the bench line times (images_per_s, trajectory_steps_per_s, single_image): the un-routed B = 4 `start` and `fusion_base` plans, the
# B = 2 `plain` plan (33 of a LoRA image's 75 calls; the one-workgroup-per-CU 64x160 / 32x160 tilings), and the 4-seed co-batched forms of all four
# (B = 16 / 8, `w_period` weight sets, their own table entries) that produce the line's images/s -- bf16 and fp8
# (8 seeds: BASELINE config 4's one-GPU share -- 64 seeds over 8 GPUs -- and bench.py's default --traj-cobatch since round 5)
_CALL_CASES = [(ck, ns) for ck in ("fusion", "fusion_base", "start", "plain") for ns in (1, 4, 8) if (ck, ns) != ("fusion", 1)]


@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("call_kind,n_seeds", _CALL_CASES)
def test_every_timed_call_kind_vs_fp32_oracle(sdxl_weights, sdxl_bundles, call_kind, n_seeds, fp8):
    """SDXL shapes at latent 128x128 (what bench.py times), LoRA concepts, the sampler's own plan builder for `call_kind` and `n_seeds` co-batched
    trajectories (rows seed-major; every seed has its OWN latent), shipped tile table asserted, hipGraph replay -- against the fp32 oracle, one oracle
    call per seed (rows of different seeds never interact).  Reference call sites: fusion_sampling.py:324-340 (fusion rows), :342-359 (start rows),
    :360-374,406,440 (B = 2 calls); the LoRA hooks route only inside the window and only at batch 4 (utils_lora.py:62), so `fusion_base`, `start`
    and `plain` run the base weights.  Bound: the one of every whole-UNet test (rel L2 <= 2e-2, max-abs <= 5e-2 max|ref|), per seed."""
    from oracle import unet_oracle as UO
    from tweediemix_amd import unet as U
    hw = 128
    plan, ehs, pooled, tid, con = _timed_plan(sdxl_weights, "lora", hw, 1, fp8=fp8, call_kind=call_kind, n_seeds=n_seeds, bundle=sdxl_bundles("lora"))
    rows = ehs.shape[0]
    routed = call_kind == "fusion"
    assert plan.B == rows * n_seeds and plan.routed == routed and plan.fp8 == fp8
    _ok, bad = U.tilings_follow_table(plan)
    deviating = [b for b in bad if b[2] is not None]     # bench.py's own assertion: a shape the shipped table holds runs the table's tiling
    assert not deviating, deviating[:5]
    g = torch.Generator().manual_seed(6)
    xs = torch.randn(n_seeds, 1, 4, hw, hw, generator=g)
    x = xs.repeat(1, rows, 1, 1, 1).reshape(n_seeds * rows, 4, hw, hw).cuda()
    eps = _graph_replay(plan, x, 601)
    orc = sdxl_bundles("lora")[2]
    worst = (0.0, 0.0)
    for s in range(n_seeds):
        sl = slice(s * rows, (s + 1) * rows)
        ref = orc.forward(x[sl], 601, ehs.cuda(), pooled.cuda(), tid.cuda(), routed=routed)
        r = rel_l2(eps[sl], ref)
        m = (eps[sl] - ref).abs().max().item() / ref.abs().max().item()
        worst = (max(worst[0], r), max(worst[1], m))
        assert torch.isfinite(eps[sl]).all() and r <= 2e-2 and m <= 5e-2, (call_kind, n_seeds, s, r, m)
    print(f"SDXL lora 1024^2 {call_kind} x {n_seeds} seed(s) (B={plan.B}) fp8={fp8}, graph replay: worst seed rel_l2={worst[0]:.4g} "
          f"max_rel={worst[1]:.4g} tilings={U.used_tilings(plan)}")


def test_plain_plan_co_batched_rows_equal_the_single_seed_call_at_sdxl_size(sdxl_weights, sdxl_bundles, monkeypatch):
    """whether attn2 runs as ONE launch (tmix_gemm_q_cross_attn: q rounded after scale * log2e, its own softmax) or as the to_q GEMM +
    attention pair must be a function of the image's shape only -- never of the batch -- or a co-batched seed stops matching its single-seed run.
    The B = 2 `plain` call at latent 128 x 128 (the 32 x 32 level's 64 tiles per image) against the same rows co-batched 2 x (B = 4), ONE tiling
    everywhere (TMIX_FORCE_TILE: the tuner's per-shape choices add LayerNorm partials in different orders): the same kernels per layer, and eps of
    the first seed's rows equal bit for bit.  Call sites: fusion_sampling.py:360-374,406,440."""
    monkeypatch.setenv("TMIX_FORCE_TILE", "1")
    hw = 128
    one, ehs, pooled, tid, _con = _timed_plan(sdxl_weights, "lora", hw, 1, call_kind="plain", n_seeds=1, bundle=sdxl_bundles("lora"))
    two, _e, _p, _t, _c = _timed_plan(sdxl_weights, "lora", hw, 1, call_kind="plain", n_seeds=2, bundle=sdxl_bundles("lora"))
    assert one.B == 2 and two.B == 4

    def names(plan):
        return [getattr(fn, "__name__", "") for fn, _a in plan.ops if getattr(fn, "__name__", "") != "tmix_gemm_prefetch_next"]
    assert names(one) == names(two), "the launch list of a call must not depend on how many seeds share it"
    assert names(one).count("tmix_gemm_q_cross_attn") == 70
    g = torch.Generator().manual_seed(6)
    xs = torch.randn(2, 1, 4, hw, hw, generator=g)
    x2 = xs.repeat(1, 2, 1, 1, 1).reshape(4, 4, hw, hw).cuda()
    e1 = one(x2[:2], 601).clone()
    e2 = two(x2, 601).clone()
    torch.cuda.synchronize()
    assert torch.isfinite(e1).all() and torch.equal(e1, e2[:2]), float((e1 - e2[:2]).abs().max())


def test_low_rank_lora_full_size_sdxl_vs_oracle(sdxl_weights):
    """`--lora_mode lowrank` at the REAL SDXL widths (2.57 B parameters, 280 routed projections each behind a tmix_lora_down launch,
    shared weights over K + 64 columns), latent 64 x 64, B = 4 concept-routed rows, the sampler's own plan builder, hipGraph replay,
    against the fp32 oracle: the bf16 bound of the merged form (rel L2 <= 2e-2)."""
    from oracle import unet_oracle as UO
    from tweediemix_amd import unet as U
    plan, ehs, pooled, tid, con = _timed_plan(sdxl_weights, "lora", 64, 1, lora_mode="lowrank")
    assert plan.lowrank and not plan.routed
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 4, 64, 64, generator=g).repeat(4, 1, 1, 1).cuda()
    eps = _graph_replay(plan, x, 601)
    ref = UO.UNetOracle(UO.SDXL, sdxl_weights, _oracle_concepts("lora", con)).forward(x, 601, ehs.cuda(), pooled.cuda(), tid.cuda(), routed=True)
    r = rel_l2(eps, ref)
    m = (eps - ref).abs().max().item() / ref.abs().max().item()
    print(f"SDXL lora (low-rank form) 512^2 B=4, one chain, graph replay: rel_l2={r:.4g} max_rel={m:.4g}")
    assert torch.isfinite(eps).all() and r <= 2e-2 and m <= 5e-2, (r, m)


def test_fp8_projections_tiny_unet_vs_oracle():
    """--dtype fp8 on the tiny UNet: attn1 q/k/v and both FF projections on e4m3 operands (tmix_gemm_fp8 behind one quantiser
    launch each), everything else bf16 -- against the fp32 oracle.  States the cost: bf16 path ~4e-3, fp8 within 3e-2 (a 3-level toy network with 64-wide
    rows has nothing to average the e4m3 rounding over; the SDXL-size test below holds the bf16 bound)."""
    from tweediemix_amd import unet as U
    orc, plan, x, ehs, pooled, tid = make("custom", 4, 16, 16, True)
    ref = orc.forward(x, 500, ehs, pooled, tid, routed=True)
    base = rel_l2(plan(x.cuda(), 500).clone().cpu(), ref)
    p8 = U.UNetPlan(plan.W, 4, 16, 16, plan.kv, pooled, tid, routed=True, fp8=True)
    got = p8(x.cuda(), 500).clone().cpu()
    r = rel_l2(got, ref)
    print(f"tiny UNet rel-L2 vs fp32 oracle: bf16 {base:.3e}, fp8 projections {r:.3e}")
    assert torch.isfinite(got).all() and base <= 2e-2 and r <= 3e-2, (base, r)


@pytest.mark.parametrize("kind,hw", [("lora", 128), ("custom", 64)])
def test_fp8_projections_full_size_sdxl_vs_oracle(sdxl_weights, sdxl_bundles, kind, hw):
    """the fp8 bench leg (`other_configs.fp8`: one chain at B = 4, latent 128 x 128, LoRA rows; and Custom-Diffusion rows at 64 x 64)
    built by the sampler's own builder, hipGraph replay, vs the fp32 oracle.  Bound: whole-UNet rel-L2 <= 2e-2 -- the SAME bound as
    the bf16 path (measured 5.4e-3 fp8 vs 5.3e-3 bf16: e4m3 operands with power-of-two block scales cost less than bf16 activations)."""
    from oracle import unet_oracle as UO
    plan, ehs, pooled, tid, con = _timed_plan(sdxl_weights, kind, hw, 1, fp8=True, bundle=sdxl_bundles(kind))
    assert plan.fp8
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 4, hw, hw, generator=g).repeat(4, 1, 1, 1).cuda()
    eps = _graph_replay(plan, x, 601)
    ref = sdxl_bundles(kind)[2].forward(x, 601, ehs.cuda(), pooled.cuda(), tid.cuda(), routed=True)
    r = rel_l2(eps, ref)
    m = (eps - ref).abs().max().item() / ref.abs().max().item()
    print(f"SDXL {kind} {hw * 8}^2 B=4 fp8 projections, one chain, graph replay: rel_l2={r:.4g} max_rel={m:.4g}")
    assert torch.isfinite(eps).all() and r <= 2e-2 and m <= 5e-2, (r, m)


# ------------------------------------------------------------------------------------------------------------------------------
# hostile activation statistics (weights.HOSTILE): what N(0, 1/fan_in) weights never show and real checkpoints are known for --
# LayerNorm rows far from zero mean, outlier channels two orders of magnitude above the rest, GroupNorm groups with large means,
# GEGLU gates deep in the erf tail.  They land in the single-pass variance of the folded LayerNorm (gemm_kernel.h ln_reduce), in the
# column partials of the GroupNorm producers, in the MX block scales of the e4m3 copies and in gelu_erf_f.
@pytest.mark.parametrize("fp8", [False, True])
@pytest.mark.parametrize("kind,routed", [("lora", True), ("custom", True), ("none", False)])
def test_tiny_unet_under_hostile_statistics(kind, routed, fp8):
    from tweediemix_amd import unet as U
    B = 4 if kind != "none" else 2
    orc, plan, x, ehs, pooled, tid = make(kind, B, 16, 16, routed, hostile=True)
    if fp8:
        plan = U.UNetPlan(plan.W, B, 16, 16, plan.kv, pooled, tid, routed=routed, fp8=True)
    ref = orc.forward(x, 601, ehs, pooled, tid, routed=routed)
    eps = plan(x.cuda(), 601).float().cpu()
    torch.cuda.synchronize()
    r = rel_l2(eps, ref)
    m = (eps - ref).abs().max().item() / ref.abs().max().item()
    cond = plan.norm_condition(x.cuda(), 601)
    worst = max(c[3] for c in cond)
    print(f"hostile tiny UNet {kind} fp8={fp8}: rel_l2={r:.4g} max_rel={m:.4g}; LayerNorm rows: max |mean|/std = {worst:.3g} over {len(cond)} sites")
    assert torch.isfinite(eps).all()
    assert worst >= 1.0, "the hostile weights must actually move the LayerNorm rows off zero mean"
    # measured 1.83e-2 - 1.97e-2 (friendly weights: ~4e-3): a 3-level network with 64- to 256-wide rows has little to average rounding noise over, and
    # a row offset of 4 std costs a bf16 stream two of its eight mantissa bits; the SDXL-width test below holds the 2e-2 of every other test
    assert r <= 3e-2 and m <= 5e-2, (r, m)


@pytest.fixture(scope="module")
def sdxl_hostile_weights():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from tweediemix_amd import unet as U, weights as Wt
    return Wt.synthetic_state_dict(U.SDXL, seed=1234, device="cuda", dtype=torch.float32, hostile=True)


@pytest.mark.parametrize("fp8", [False, True])
def test_sdxl_width_unet_under_hostile_statistics(sdxl_hostile_weights, fp8):
    """the real SDXL widths (1280-wide rows, 10-layer transformers, 2.57 B parameters) at latent 64 x 64, B = 4 LoRA-routed rows, the
    sampler's own plan builder, hipGraph replay -- with the hostile biases -- against the fp32 oracle at the bound of every other
    whole-UNet test (rel L2 <= 2e-2, bf16 and fp8 plans alike), plus the calibration readout of the LayerNorm rows."""
    from oracle import unet_oracle as UO
    from tweediemix_amd import unet as U
    plan, ehs, pooled, tid, con = _timed_plan(sdxl_hostile_weights, "lora", 64, 1, fp8=fp8)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 4, 64, 64, generator=g).repeat(4, 1, 1, 1).cuda()
    eps = _graph_replay(plan, x, 601)
    ref = UO.UNetOracle(UO.SDXL, sdxl_hostile_weights, _oracle_concepts("lora", con)).forward(x, 601, ehs.cuda(), pooled.cuda(), tid.cuda(), routed=True)
    r = rel_l2(eps, ref)
    m = (eps - ref).abs().max().item() / ref.abs().max().item()
    cond = plan.norm_condition(x, 601)
    worst = max(c[3] for c in cond)
    print(f"hostile SDXL-width UNet 512^2 B=4 fp8={fp8}: rel_l2={r:.4g} max_rel={m:.4g}; LayerNorm rows: max |mean|/std = {worst:.3g} "
          f"over {len(cond)} sites ({sum(c[3] > U.LN_COND_WARN for c in cond)} beyond LN_COND_WARN)")
    assert torch.isfinite(eps).all() and worst >= 1.0
    assert r <= 2e-2 and m <= 5e-2, (r, m)


def test_real_checkpoint_activation_statistics_if_available():
    """every parity test above uses random-init weights; a real SDXL checkpoint has very different activation statistics
    (large LayerNorm / GroupNorm means, outlier channels) for the bf16 single-pass sum / sum-of-squares norms.  No checkpoint
    exists offline: point TMIX_REAL_SDXL_UNET at a diffusers-format UNet weights file (.safetensors / .bin) to run it."""
    import os
    path = os.environ.get("TMIX_REAL_SDXL_UNET")
    if not path or not os.path.exists(path):
        pytest.skip("set TMIX_REAL_SDXL_UNET=<unet/diffusion_pytorch_model.safetensors> to run the real-weights parity check")
    from oracle import unet_oracle as UO
    from tweediemix_amd import unet as U
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu")
    sd = {k: v.float().cuda() for k, v in sd.items()}
    g = torch.Generator().manual_seed(9)
    B, hw = 2, 64
    ehs = torch.randn(B, 77, U.SDXL.cross_dim, generator=g).to(torch.bfloat16).float()
    pooled = torch.randn(B, U.SDXL.pooled_dim, generator=g)
    tid = torch.tensor([[512.0, 512, 0, 0, 512, 512]] * B)
    x = torch.randn(1, 4, hw, hw, generator=g).repeat(B, 1, 1, 1).cuda()
    W = U.UNetWeights(U.SDXL, sd, "cuda", None)
    plan = U.UNetPlan(W, B, hw, hw, U.KVCache(W, ehs, [0] * B), pooled, tid)
    eps = plan(x, 601).clone()
    ref = UO.UNetOracle(UO.SDXL, sd).forward(x, 601, ehs.cuda(), pooled.cuda(), tid.cuda())
    r = rel_l2(eps, ref)
    print(f"real SDXL weights 512^2 B=2: rel_l2={r:.4g}")
    assert torch.isfinite(eps).all() and r <= 2e-2, r
