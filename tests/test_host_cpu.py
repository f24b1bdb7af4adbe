"""CPU: host-side logic that needs no GPU -- schedule tables and masks of the product package against the golden
vectors, parameter inventories, CLI flag surface, tile/plan bookkeeping."""
import os

import pytest
import numpy as np
import torch


def test_product_schedule_matches_golden(golden_dir):
    from tweediemix_amd.schedule import Schedule
    g = np.load(os.path.join(golden_dir, "schedule.npz"))
    for n in (20, 50):
        s = Schedule(n)
        assert s.timesteps == [int(v) for v in g[f"timesteps_{n}"]]
        assert all(np.float32(s.alpha(t)) == a for t, a in zip(s.timesteps, g[f"alpha_{n}"]))
        assert all(np.float32(s.alpha(t - s.skip)) == a for t, a in zip(s.timesteps, g[f"alpha_next_{n}"]))
    assert Schedule(50).alpha(-19) == Schedule(50).final_alpha_cumprod


def test_product_masks_match_golden(golden_dir):
    from tweediemix_amd import masks as M
    g = np.load(os.path.join(golden_dir, "masks.npz"))
    for hw in (128, 64):
        m = M.build_masks([g["img_a_cat"], g["img_a_dog"]], hw, hw, device="cpu")
        assert np.array_equal(m.numpy(), g[f"masks_all_{hw}"])
    r = M.random_rectangle_masks(3, 256, 256, seed=1)
    assert len(r) == 2 and all(0.08 < (a > 0).mean() < 0.35 for a in r)


def test_parameter_inventories():
    from tweediemix_amd import unet as U, vae as V, weights as Wt
    n = sum(int(np.prod(s)) for s in Wt.param_shapes(U.SDXL).values())
    assert n == 2_567_463_684                                  # SDXL-base UNet
    assert len(U.attention_blocks(U.SDXL)) == 70
    assert sum(int(np.prod(s)) for s in V.param_shapes(V.FULL).values()) == 49_490_199
    sd = Wt.synthetic_state_dict(U.TINY)
    assert set(sd) == set(Wt.param_shapes(U.TINY))
    c = Wt.synthetic_concepts(U.TINY, "lora", 2)
    assert all(v.shape[0] == 4 or v.shape[1] == 4 for v in c[0].values())     # rank 4 (model_lora.py:28-48)


def test_cli_flag_surface():
    """every flag of the reference's argparse block (fusion_sampling.py:534-585, + --t_stop in the LoRA script)"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fs_cli_cpu", os.path.join(root, "fusion_generation", "fusion_sampling.py"))
    fs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fs)
    ref_flags = ["seed", "device", "output_path", "output_path_all", "negative_prompt", "sd_version", "t_cond", "guidance_scale",
                 "n_timesteps", "prompt", "prompt_orig", "seg_concepts", "personal_checkpoint", "concepts", "modifier_token",
                 "resampling_steps", "jumping_steps", "seg_gpu", "crops_coords_top_left_h", "crops_coords_top_left_w",
                 "resolution_h", "resolution_w"]
    opt = fs.build_parser().parse_args([])
    for f in ref_flags:
        assert hasattr(opt, f), f
    assert (opt.seed, opt.t_cond, opt.guidance_scale, opt.n_timesteps, opt.resampling_steps, opt.jumping_steps) == (182, 0.4, 9.0, 50, 10, 5)
    fs.LORA = True
    assert fs.build_parser().parse_args([]).t_stop == 0.9
    fs.LORA = False
    # additive flags: default arithmetic is bf16, e4m3 operands only on request
    assert opt.dtype == 'bf16' and fs.build_parser().parse_args(['--dtype', 'fp8']).dtype == 'fp8'
    with pytest.raises(SystemExit):
        fs.build_parser().parse_args(['--dtype', 'fp16'])


def test_geglu_interleave_is_a_permutation():
    from tweediemix_amd.weights import interleave_geglu
    w = torch.arange(64 * 4, dtype=torch.float32).reshape(64, 4)
    wi, _ = interleave_geglu(w, None)
    assert sorted(wi[:, 0].tolist()) == w[:, 0].tolist()
    assert wi[0, 0] == w[0, 0] and wi[16, 0] == w[32, 0] and wi[32, 0] == w[16, 0]     # [value 0-15 | gate 0-15 | value 16-31 ...]


def test_expand_masks_matches_reference_sidecar(golden_dir):
    """bounding-rectangle + overlap rule of text_segment/run_expand.py:35-87, against what the reference script
    itself saved (oracle/gen_golden_masks.py ran it with prepared SAM masks)."""
    from oracle import tweedie_oracle as TO
    from tweediemix_amd import masks as M
    g = np.load(os.path.join(golden_dir, "expand_masks.npz"))
    for case in ("disjoint", "overlap", "contained", "touching"):
        ins = [g[f"{case}_in0"], g[f"{case}_in1"]]
        for impl in (M.expand_masks, TO.expand_masks):
            out = impl(ins)
            for i in range(2):
                assert np.array_equal(out[i], g[f"{case}_out{i}"].astype(bool)), (case, i, impl.__module__)
    # >80 % rule fired in the 'contained' case: mask 1 lost the overlap box, mask 0 kept only its original pixels there
    assert g["contained_out0"].sum() == g["contained_in0"].sum()


def test_clip_bpe_tokenizer_matches_transformers_vectors(golden_dir):
    """tweediemix_amd/text.py ClipBPETokenizer against ids produced by transformers' CLIPTokenizer on the synthetic
    vocabulary of tests/golden/clip_tok (oracle/gen_golden_tokenizer.py): case folding, whitespace, contractions,
    digits, non-ASCII bytes, truncation to 77, both pad conventions, added modifier tokens."""
    import json, os
    from tweediemix_amd.text import ClipBPETokenizer
    d = os.path.join(golden_dir, "clip_tok")
    cases = json.load(open(os.path.join(d, "cases.json")))
    vocab = json.load(open(os.path.join(d, "vocab.json")))
    for c in cases:
        tok = ClipBPETokenizer.from_pretrained(d)
        tok.pad_token = c["pad_token"]
        assert tok(c["texts"]).tolist() == c["ids_plain"]
        n0 = len(tok)
        assert n0 == len(vocab)
        for t, want in zip(c["added"], c["added_ids"]):
            assert tok.add_tokens(t) == 1 and tok.convert_tokens_to_ids(t) == want
        assert tok.add_tokens(c["added"][0]) == 0 and len(tok) == c["len_after"]
        assert [tok.tokenize(t) for t in c["texts"]] == c["tokens_added"]
        ids = tok(c["texts"])
        assert ids.shape == (len(c["texts"]), 77) and ids.tolist() == c["ids_added"]


def test_prompt_assembly_and_token_injection_match_reference_vectors(golden_dir):
    """text.assemble_prompts / inject_modifier_tokens against tests/golden/prompts.json, produced by executing the
    reference's own statements (oracle/gen_golden_prompts.py): modifier token placement incl. the find() == -1 case,
    prompts_single, new token ids in both vocabularies and the embedding rows they receive."""
    import json, os, types
    import torch
    from tweediemix_amd import text as T

    class Enc:                                   # the two methods inject_modifier_tokens uses of ClipTextEncoder
        def __init__(self, n, d):
            self.tok = torch.zeros(n, d)
        resize_token_embeddings = T.ClipTextEncoder.resize_token_embeddings
        set_token_embedding = T.ClipTextEncoder.set_token_embedding
        dev, d = torch.device("cpu"), None

    for c in json.load(open(os.path.join(golden_dir, "prompts.json"))):
        a = c["args"]
        prompts, single, K = T.assemble_prompts(a["prompt"], a["prompt_orig"], a["concepts"], a["modifier_token"])
        assert prompts == c["prompts"] and single == c["prompts_single"] and K == c["concept_num"]
        sts = [{"modifier_token": {name: torch.tensor(v1)}, "modifier_token_2": {name: torch.tensor(v2)}} for name, v1, v2 in c["ckpt_tokens"]]
        toks = [T.ClipBPETokenizer({f"t{i}": i for i in range(50)}, []), T.ClipBPETokenizer({f"t{i}": i for i in range(60)}, [])]
        encs = [Enc(50, 8), Enc(60, 12)]
        encs[0].d, encs[1].d = 8, 12
        ids, ids_2 = T.inject_modifier_tokens(toks, encs, sts, a["modifier_token"].split('+'))
        assert ids == c["ids"] and ids_2 == c["ids_2"]
        torch.testing.assert_close(encs[0].tok[50:], torch.tensor(c["table_rows"]))
        torch.testing.assert_close(encs[1].tok[60:], torch.tensor(c["table_rows_2"]))
    assert T.inject_modifier_tokens(toks, encs, [{"unet": {}}], ["<x>"]) == ([], [])        # no 'modifier_token' in sts[0]: untouched


def test_video_image_preprocessing_matches_reference_vectors(golden_dir):
    """video.center_crop_wide / resize_bilinear / prepare_image_latents against outputs of the reference's own functions
    (tests/golden/video_image.npz, oracle/gen_golden_video.py)."""
    import os
    import numpy as np
    import torch
    from PIL import Image
    from tweediemix_amd import video as V
    z = np.load(os.path.join(golden_dir, "video_image.npz"))
    img = Image.fromarray(z["img"])
    assert np.array_equal(np.array(V.center_crop_wide(img, (64, 64))), z["crop.sq"])
    assert np.array_equal(np.array(V.center_crop_wide(img, (96, 56))), z["crop.wide"])
    assert np.array_equal(np.array(V.resize_bilinear(V.center_crop_wide(img, (64, 64)), (32, 32))), z["resize.224"])
    got = V.prepare_image_latents(torch.from_numpy(z["pil.mean"]), 16)
    assert np.array_equal(got.numpy(), z["pil.out"])
    pv = V.clip_pixel_values(img)
    assert pv.shape == (1, 3, 90, 150) and abs(float(pv.mean())) < 3


def test_tile_table_contexts():
    """a chain that shares the chip reads its own entry, falls back to the plain one, and never inherits a
    one-workgroup-per-CU tiling; the shipped table holds both contexts for the headline launch shapes."""
    from tweediemix_amd import unet as U, lib as L
    saved = dict(U._TUNE_CACHE)
    try:
        U._TUNE_CACHE.clear()
        U._TUNE_CACHE.update({"a": 3, U.SHARED + "a": 2, "b": 7, "c": L.TILE_EXCLUSIVE[0]})
        assert U.tune_lookup("", "a") == 3 and U.tune_lookup(U.SHARED, "a") == 2
        assert U.tune_lookup("", "b") == 7 and U.tune_lookup(U.SHARED, "b") == 7
        assert U.tune_lookup("", "c") == L.TILE_EXCLUSIVE[0] and U.tune_lookup(U.SHARED, "c") is None
        assert U.tune_lookup("", "missing") is None and U.tune_lookup(U.SHARED, "missing") is None
    finally:
        U._TUNE_CACHE.clear()
        U._TUNE_CACHE.update(saved)
    ff1 = "('gemm', 2048, 10240, 1280, 1, 1, False, False, False, False, True)"      # GEGLU up-projection of a two-row chain
    assert U.tune_lookup("", ff1) in L.TILE_CANDIDATES and U.tune_lookup(U.SHARED, ff1) in L.TILE_CANDIDATES
    assert all(1 <= v <= L.TILE_COUNT for v in U._TUNE_CACHE.values())
    # every one-workgroup-per-CU tiling is a valid id; the ones the tuner may pick are a subset (24 is by request only)
    assert all(1 <= t <= L.TILE_COUNT for t in L.TILE_EXCLUSIVE)
    assert set(L.TILE_EXCLUSIVE) - {24} <= set(L.TILE_CANDIDATES) and 24 not in L.TILE_CANDIDATES



def test_video_alphas_follow_the_scheduler_config():
    """run_video.py builds alphas_cumprod from the checkpoint's scheduler_config.json the way diffusers' DDIMScheduler does
    (float32 tensors): scaled_linear reproduces the image sampler's table (tests/golden/schedule.npz, produced from the
    reference's own scheduler arithmetic); zero-terminal-SNR rescaling ends at exactly 0; the keyword arguments follow the file."""
    import os
    from tweediemix_amd import video as V
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schedule.npz"))
    acp, kw = V.alphas_from_scheduler_config(dict(beta_schedule="scaled_linear", beta_start=0.00085, beta_end=0.012, steps_offset=1,
                                                  set_alpha_to_one=False))
    key = [k for k in g.files if "alphas_cumprod" in k][0]
    ref = np.asarray(g[key], np.float32)
    ref = ref[1:] if len(ref) == 1001 else ref             # the image sampler's table has 1.0 prepended (fusion_sampling.py:218)
    assert np.array_equal(acp, ref)
    assert kw == dict(steps_offset=1, set_alpha_to_one=False)
    acp0, kw0 = V.alphas_from_scheduler_config(dict(beta_schedule="squaredcos_cap_v2", rescale_betas_zero_snr=True))
    assert acp0[-1] == 0.0 and 0.999 < acp0[0] < 1.0 and np.all(np.diff(acp0) <= 0) and kw0 == dict(steps_offset=0, set_alpha_to_one=True)
    sch = V.VideoSchedule(acp0, 50, **kw0)
    assert sch.final_alpha_cumprod == 1.0 and sch.timesteps[0] == 980 and sch.timesteps[-1] == 0


def test_weights_tolerate_checkpoints_without_attn1_lora_or_custom_kv():
    """the reference trainer's freeze_model='lora' checkpoints hold attn2 pairs only, and a Custom-Diffusion delta may not cover
    every block (fusion_sampling.py:206-209 `if name in single_st['unet']`): a missing pair is a zero delta / the base K,V."""
    from tweediemix_amd import unet as U, weights as Wt
    cfg = U.TINY
    sd = Wt.synthetic_state_dict(cfg, seed=3, device="cpu", dtype=torch.float32)
    tb0 = U.attention_blocks(cfg)[0][0]
    con = Wt.synthetic_concepts(cfg, "lora", 2, device="cpu")
    con = [{k: v for k, v in c.items() if ".attn1." not in k} for c in con]
    W = U.UNetWeights(cfg, sd, "cpu", ("lora", con))
    a1 = tb0 + ".attn1"
    rows = W[a1 + ".out_rows"]
    assert rows.shape[0] == 3 and torch.equal(rows[1], rows[0]) and torch.equal(rows[2], rows[0])        # zero delta
    a2 = tb0 + ".attn2"
    assert not torch.equal(W[a2 + ".out_rows"][1], W[a2 + ".out_rows"][0])                               # attn2 pair applied
    cc = Wt.synthetic_concepts(cfg, "custom", 2, device="cpu")
    cc[1] = {k: v for k, v in cc[1].items() if not k.startswith(tb0)}
    Wc = U.UNetWeights(cfg, sd, "cpu", ("custom", cc))
    kv = Wc[a2 + ".kv_rows"]
    assert kv.shape[0] == 3 and torch.equal(kv[2], kv[0]) and not torch.equal(kv[1], kv[0])


def test_stacked_time_projection_sections_cover_every_resnet():
    """UNetWeights.stacked_time_proj (one tmix_linear_small_sections launch for all ResnetBlock2D.time_emb_proj layers): the stack
    is the per-resnet matrices in a fixed order, the section table is their running row count, and every resnet has a section."""
    from tweediemix_amd import unet as U, weights as Wt
    cfg = U.TINY
    sd = Wt.synthetic_state_dict(cfg, seed=5, device="cpu", dtype=torch.float32)
    W = U.UNetWeights(cfg, sd, "cpu")
    w, b, starts, names = W.stacked_time_proj()
    assert names == sorted(k[:-len(".time_emb_proj.weight")] for k in sd if k.endswith(".time_emb_proj.weight"))
    st = starts.tolist()
    assert st[0] == 0 and st[-1] == w.shape[0] == b.shape[0] and len(st) == len(names) + 1
    for i, n in enumerate(names):
        assert torch.equal(w[st[i]:st[i + 1]], W[n + ".time_emb_proj.weight"])
        assert torch.equal(b[st[i]:st[i + 1]], W[n + ".time_emb_proj.bias"].float())
    assert W.stacked_time_proj() is W.stacked_time_proj()          # built once per checkpoint


def test_f8copy_layout():
    """ops.F8Copy: e4m3 bytes [rows, N] first, the MX block scales [N/32, rows] at a 256-byte aligned offset behind them -- the
    (Ct, ldct, strideCt) triple TMIX_F8_COPY_OUT reads from the descriptor."""
    from tweediemix_amd import ops, lib as L
    rows, N = 96, 160
    cp = ops.F8Copy(rows, N, "cpu")
    assert cp.off % 256 == 0 and cp.off >= rows * N and cp.nbytes == ops.F8Copy.bytes_for(rows, N) == cp.off + (N // 32) * rows
    assert cp.q.shape == (rows, N) and cp.scales.shape == (N // 32, rows)
    assert cp.q.data_ptr() == cp.buf.data_ptr() and cp.scales.data_ptr() == cp.buf.data_ptr() + cp.off
    d = L.GemmDesc()
    d.batch, d.M, d.N = 2, rows // 2, N
    cp.attach(d)
    assert d.Ct == cp.buf.data_ptr() and d.ldct == N and d.strideCt == cp.off and d.reserved0 == L.F8_COPY_OUT


def test_sidecar_directories_do_not_collide_across_ranks():
    """ranks sample different seeds concurrently: each runs its segmentation side-car in its own directory and not on a GPU
    another rank samples on; a single process keeps the reference's paths (fusion_sampling.py:453-466)."""
    from tweediemix_amd import masks as M
    assert M.sidecar_layout("out", 0, 1, 0, 1) == ("out", 1)
    dirs = [M.sidecar_layout("out", r, 4, r, 1) for r in range(4)]
    assert len({d for d, _g in dirs}) == 4 and all(d.startswith("out") for d, _g in dirs)
    assert [g for _d, g in dirs] == [0, 1, 2, 3]                    # --seg_gpu 1 is rank 1's GPU: every rank uses its own instead
    assert M.sidecar_layout("out", 2, 4, 2, 6) == ("out/rank2", 6)  # a GPU outside the sampling ranks is honoured


def test_sidecar_gpu_is_a_physical_id_when_the_parent_restricts_the_visible_devices(monkeypatch):
    """the side-car's command line sets CUDA_VISIBLE_DEVICES itself (fusion_sampling.py:458): rank-local GPU i of a job started under
    HIP_VISIBLE_DEVICES=4,5,6,7 is physical GPU 4 + i."""
    from tweediemix_amd import masks as M
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")
    assert [M.sidecar_layout("out", r, 4, r, 1)[1] for r in range(4)] == [4, 5, 6, 7]
    assert M.sidecar_layout("out", 1, 4, 1, 9) == ("out/rank1", 9)      # an explicit GPU outside the job is passed through
    assert M.sidecar_layout("out", 0, 1, 0, 1) == ("out", 1)            # one process: the reference's own arguments
    # ... and the child must not inherit HIP_VISIBLE_DEVICES (the HIP runtime would prefer it over the command line's CUDA_VISIBLE_DEVICES)
    assert "HIP_VISIBLE_DEVICES" not in M.sidecar_child_env()
    # CUDA_VISIBLE_DEVICES in the parent: replaced by the command line's own -> the parent's i-th entry
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "2,3")
    assert [M.sidecar_layout("out", r, 2, r, 1)[1] for r in range(2)] == [2, 3]
    # ROCR_VISIBLE_DEVICES only: HIP / CUDA ordinals index the filtered list -> the rank-local index
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4,5,6,7")
    assert [M.sidecar_layout("out", r, 4, r, 1)[1] for r in range(4)] == [0, 1, 2, 3]
    assert M.sidecar_child_env().get("ROCR_VISIBLE_DEVICES") == "4,5,6,7"


def test_gather_without_a_process_group_is_an_error_when_there_are_other_ranks():
    import pytest
    import torch
    from tweediemix_amd import dist as D
    x = torch.zeros(2, 4, 8, 8)
    assert D.gather_latents(x, 2, 0, 1) is x
    with pytest.raises(AssertionError):
        D.gather_latents(x, 4, 0, 2)


def test_partition_masks_are_a_convex_blend_and_keep_the_oracle_latent_bounded():
    """masks.partition_rectangle_masks: fg_1 + fg_2 + bg == 1 on every latent pixel (the blend of fusion_sampling.py:466-469 is then a convex
    combination), whereas random_rectangle_masks can intersect (weights sum to 2: the reference does not normalise, the region doubles every
    fusion step).  The second half replays the oracle's whole sampler on a stand-in eps = sqrt(1 - alpha_t) x (the ideal denoiser of unit-variance data): max|x| stays O(10)
    with partition masks and explodes with the overlapping set -- which is why bench.py and the trajectory tolerance default to the former."""
    from oracle import tweedie_oracle as TO
    from tweediemix_amd import masks as M
    K = 3
    some_overlap = False
    for res, hw in ((1024, 128), (512, 64)):
        for seed in range(12):
            m = M.build_masks(M.partition_rectangle_masks(K, res, res, seed), hw, hw, device="cpu")
            s = m.sum(0)
            assert m.shape == (K, 1, hw, hw) and float(s.min()) == 1.0 and float(s.max()) == 1.0
            assert all(float(m[c].mean()) >= 0.04 for c in range(K - 1)), "a foreground region must survive the carve-out"
            so = M.build_masks(M.random_rectangle_masks(K, res, res, seed), hw, hw, device="cpu").sum(0)
            some_overlap |= float(so.max()) == 2.0
    assert some_overlap
    assert M.synthetic_masks("partition", K, 64, 64, 1)[0].dtype == np.uint8
    with pytest.raises(ValueError):
        M.synthetic_masks("nope", K, 64, 64)

    seed = next(s for s in range(64) if float((M.build_masks(M.random_rectangle_masks(K, 256, 256, s), 32, 32, device="cpu").sum(0) > 1).float().mean()) > 0.03)
    rowscale = (1 + 0.02 * np.arange(4, dtype=np.float32))[:, None, None, None]     # stand-in eps = E[noise | x_t] of unit-variance data, slightly different per prompt row

    def run(kind):
        imgs = M.synthetic_masks(kind, K, 256, 256, seed=seed)
        o = TO.TweedieOracle(K, 20, g=0.8, t_cond=0.2, t_stop=0.8, resampling_steps=2, jumping_steps=2, mask_fn=lambda: TO.build_masks(imgs, 32, 32))
        x = np.random.RandomState(0).randn(1, 4, 32, 32).astype(np.float32)
        for t in o.sch.timesteps:
            x = o.denoise_step(x, int(t), lambda xin, t_, rows, k, routed: (np.sqrt(1 - o.sch.alpha(t_)) * xin * rowscale[:xin.shape[0]]).astype(np.float32))
        return float(np.abs(x).max())
    bounded, exploding = run("partition"), run("overlap")
    assert bounded < 10.0 and exploding > 20 * bounded, (bounded, exploding)          # 13 fusion steps at weight 2 on the overlap



def test_weight_hint_policy_caps_large_tensors_of_single_seed_calls_only(monkeypatch):
    """UNetPlan's weight hints: single-seed calls name 8 MB of tensors over 20 MB, co-batched calls whole tensors; the env knobs override both."""
    from tweediemix_amd import unet as U
    monkeypatch.delenv("TMIX_PF_CAP_MB", raising=False); monkeypatch.delenv("TMIX_PF_CAP_OVER_MB", raising=False)
    MB = 1 << 20
    cap, over = U.hint_policy(4, 128, 128)
    assert (cap, over) == (8 * MB, 20 * MB)
    assert U.hint_bytes(39 * MB, cap, over) == 8 * MB            # four merged q/k/v weight sets
    assert U.hint_bytes(26 * MB + 1, cap, over) == 8 * MB        # FF1
    assert U.hint_bytes(13 * MB, cap, over) == 13 * MB           # FF2 / routed out-projections: whole (capping these measured 0.5 - 1 ms slower)
    assert U.hint_bytes(20 * MB, cap, over) == 20 * MB
    cap, over = U.hint_policy(32, 128, 128)                      # 8 co-batched seeds
    assert cap == 0 and U.hint_bytes(39 * MB, cap, over) == 39 * MB
    assert U.hint_policy(2, 128, 128)[0] == 8 * MB and U.hint_policy(16, 128, 128)[0] == 0
    monkeypatch.setenv("TMIX_PF_CAP_MB", "0")
    assert U.hint_policy(4, 128, 128)[0] == 0
    monkeypatch.setenv("TMIX_PF_CAP_MB", "1.5"); monkeypatch.setenv("TMIX_PF_CAP_OVER_MB", "12")
    cap, over = U.hint_policy(32, 128, 128)
    assert (cap, over) == (3 * MB // 2, 12 * MB) and U.hint_bytes(13 * MB, cap, over) == 3 * MB // 2
