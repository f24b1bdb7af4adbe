"""CPU: the oracle restatement must reproduce the fixtures generated from the reference's own code
(oracle/gen_golden.py).  fp32 fixtures: <=1e-5 abs (same algorithm, fp32 both sides);
fp16-eps fixtures: same bound, with the oracle emulating every fp16 rounding point of the autocast path."""
import os

import numpy as np
import pytest

from oracle import tweedie_oracle as O

TRAJ = ["traj_custom_n50_f32", "traj_custom_n50_f16", "traj_custom_n20_f32", "traj_lora_n50_f32",
        "traj_lora_n50_f16", "traj_custom_K2_n20_f32", "traj_custom_K4_n20_f32"]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_schedule(golden_dir):
    g = load(golden_dir, "schedule")
    np.testing.assert_allclose(O.alphas_cumprod_sdxl(), g["alphas_cumprod"], rtol=0, atol=0)
    for n in (20, 50):
        s = O.Schedule(n)
        assert (s.timesteps == g[f"timesteps_{n}"]).all()
        a = np.array([s.alpha(int(t)) for t in s.timesteps], np.float32)
        an = np.array([s.alpha(int(t) - s.skip) for t in s.timesteps], np.float32)
        assert (a == g[f"alpha_{n}"]).all() and (an == g[f"alpha_next_{n}"]).all()
    assert O.Schedule(50).timesteps[0] == 981 and O.Schedule(20).timesteps[0] == 951


def test_masks(golden_dir):
    g = load(golden_dir, "masks")
    for hw in (128, 64):
        fgs = [g["img_a_cat"], g["img_a_dog"]]
        for nm, im in zip(("a_cat", "a_dog"), fgs):
            assert (O.preprocess_mask(im, hw, hw) == g[f"mask_{nm}_{hw}"]).all()
        m = O.build_masks(fgs, hw, hw)
        assert (m == g[f"masks_all_{hw}"]).all()
    assert int(g["masks_all_64"][0].sum()) == 595 and int(g["masks_all_64"][1].sum()) == 828


def replay(g, lowp):
    K, n = int(g["K"]), int(g["n"])
    h, w = int(g["h"]), int(g["w"])
    lora = bool(g["lora"])
    o = O.TweedieOracle(K, n, g=float(g["guidance_scale"]), t_cond=float(g["t_cond"]),
                        t_stop=float(g["t_stop"]) if lora else None,
                        resampling_steps=int(g["resampling_steps"]), jumping_steps=int(g["jumping_steps"]),
                        lowp=lowp, mask_fn=lambda: O.build_masks(list(g["mask_images"]), h, w))
    idx = [0]
    rowval = {"e": 0.0, "s": 100.0}

    def unet_fn(x, t, rows, kind, routed):
        i = idx[0]
        idx[0] += 1
        assert x.shape[0] == int(g["req_B"][i]) and t == int(g["req_t"][i]), (i, x.shape, t)
        ids = np.array([rowval[k] + r for k, r in rows], np.float32)
        assert (ids == g[f"req{i}_rows"]).all() and (ids == g[f"req{i}_prow"]).all(), (i, ids)
        tol = 2e-5
        np.testing.assert_allclose(x, g[f"req{i}_x"], rtol=tol, atol=tol)
        in_window = t in set(int(v) for v in g["t_cond_list"])
        assert routed == (in_window and x.shape[0] == 4)
        return g[f"req{i}_eps"]

    x = g["xs"][0]
    outs = [x]
    for t in g["timesteps"]:
        x = o.denoise_step(x, int(t), unet_fn)
        outs.append(x)
    assert idx[0] == len(g["req_B"])
    return np.stack(outs), o


@pytest.mark.parametrize("name", TRAJ)
def test_trajectory(golden_dir, name):
    g = load(golden_dir, name)
    lowp = np.float16 if int(g["eps_is_fp16"]) else None
    O.CPU_SCALAR_TENSOR_SEMANTICS = True      # fixtures were produced by torch-CPU (see oracle docstring)
    try:
        xs, o = replay(g, lowp)
    finally:
        O.CPU_SCALAR_TENSOR_SEMANTICS = False
    tol = 2e-5
    np.testing.assert_allclose(xs, g["xs"], rtol=tol, atol=tol)
    assert (o.masks == g["masks"]).all()
    assert int(g["n_sys_calls"]) == 1


def test_call_schedule_default_flags(golden_dir):
    g = load(golden_dir, "traj_custom_n50_f32")
    assert int((g["req_B"] == 4).sum()) == 51 and int((g["req_B"] == 2).sum()) == 24
    g = load(golden_dir, "traj_lora_n50_f32")
    assert int((g["req_B"] == 4).sum()) == 42 and int((g["req_B"] == 2).sum()) == 33


def _lora_of(g, which, K=3):
    return [{nm: (g[f"lora_{which}_{nm}{i}_down"], g[f"lora_{which}_{nm}{i}_up"]) for nm in ("q", "k", "v", "out")}
            for i in range(K)]


@pytest.mark.parametrize("B", [4, 2])
@pytest.mark.parametrize("tag", ["in", "out"])
def test_sa_forward_custom(golden_dir, B, tag):
    g = load(golden_dir, "attention")
    p = "custom_attn2"
    y = O.sa_forward_custom(g[f"{p}_B{B}_x"], g[f"{p}_B{B}_ehs"], g[f"{p}_wq"], g[f"{p}_wk"], g[f"{p}_wv"],
                            g[f"{p}_wo"], g[f"{p}_bo"], [g[f"{p}_wk{i}"] for i in range(3)],
                            [g[f"{p}_wv{i}"] for i in range(3)], heads=2, scale=64 ** -0.5, routed=(tag == "in"))
    np.testing.assert_allclose(y, g[f"{p}_B{B}_{tag}_y"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("which", ["attn1", "attn2"])
@pytest.mark.parametrize("B", [4, 2])
@pytest.mark.parametrize("tag", ["in", "out"])
def test_sa_forward_lora(golden_dir, which, B, tag):
    g = load(golden_dir, "attention")
    p = f"lora_{which}"
    ehs = g[f"{p}_B{B}_ehs"] if which == "attn2" else None
    y = O.sa_forward_lora(g[f"{p}_B{B}_x"], ehs, g[f"{p}_wq"], g[f"{p}_wk"], g[f"{p}_wv"], g[f"{p}_wo"],
                          g[f"{p}_bo"], _lora_of(g, which), heads=2, scale=64 ** -0.5, routed=(tag == "in"))
    np.testing.assert_allclose(y, g[f"{p}_B{B}_{tag}_y"], rtol=1e-4, atol=1e-4)


def test_lora_merged_equivalence(golden_dir):
    """row i+1 of the routed batch == plain attention with W + up@down (what the HIP path runs)."""
    g = load(golden_dir, "attention")
    p = "lora_attn1"
    x = g[f"{p}_B4_x"]
    lo = _lora_of(g, "attn1")
    for i in range(3):
        mw = {nm: O.lora_merged_weight(g[f"{p}_w{nm[0] if nm != 'out' else 'o'}"], *lo[i][nm]) for nm in ("q", "k", "v", "out")}
        y = O.sa_forward_lora(x[i + 1:i + 2], None, mw["q"], mw["k"], mw["v"], mw["out"], g[f"{p}_bo"], [],
                              heads=2, scale=64 ** -0.5, routed=False)
        np.testing.assert_allclose(y[0], g[f"{p}_B4_in_y"][i + 1], rtol=1e-4, atol=1e-4)


def _clip_case(name, golden_dir):
    import torch
    z = np.load(os.path.join(golden_dir, "clip_text.npz"))
    src = "g" if name == "e" else name
    sd = {k[len(src) + 4:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith(src + ".sd.")}
    heads, eos, gelu, _n = [int(v) for v in z[name + ".meta"]]
    return z, sd, heads, eos, ("gelu" if gelu else "quick_gelu")


@pytest.mark.parametrize("name", ["l", "g", "e"])
def test_clip_oracle_matches_transformers_vectors(name, golden_dir):
    """oracle/clip_oracle.py against outputs of the transformers CLIP text models (tests/golden/clip_text.npz):
    hidden_states[-2], final-norm output and the pooled / projected row for both pooled-position rules."""
    import torch
    from oracle import clip_oracle as CO
    z, sd, heads, eos, act = _clip_case(name, golden_dir)
    ids = torch.from_numpy(z[name + ".ids"])
    o = CO.clip_text_forward(sd, ids, heads, act, eos_token_id=eos)
    torch.testing.assert_close(o["hidden_states"][-2], torch.from_numpy(z[name + ".hs_m2"]), rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(o["last_hidden_state"], torch.from_numpy(z[name + ".last"]), rtol=1e-4, atol=2e-5)
    pooled = o["text_embeds"] if o["text_embeds"] is not None else o["pooler_output"]
    torch.testing.assert_close(pooled, torch.from_numpy(z[name + ".pooled"]), rtol=1e-4, atol=2e-5)
    if name == "g":          # legacy eos_token_id == 2: the pooled row sits at the largest id (the added token), not at EOS
        assert int(ids[1].argmax()) == 3 and int(ids[2].argmax()) == 7


def test_video_step_and_injection_match_reference_vectors(golden_dir):
    """oracle video_vpred_step / inject_first_frame against tests/golden/video_step.npz, produced by executing the
    reference's statements (oracle/gen_golden_video.py): fp32 to 1e-6, fp16 bit-exact with torch-CPU scalar semantics."""
    z = np.load(os.path.join(golden_dir, "video_step.npz"))
    acp = z["alphas_cumprod"]
    for dt, lowp in (("f32", None), ("f16", np.float16)):
        O.CPU_SCALAR_TENSOR_SEMANTICS = lowp is not None
        try:
            for case in range(4):
                k = f"step.{dt}.{case}"
                t, skip, g = z[k + ".meta"]
                at, atn = O.video_alpha(acp, acp[0], int(t)), O.video_alpha(acp, acp[0], int(t) - int(skip))
                got = O.video_vpred_step(z[k + ".x"], z[k + ".v"], g, at, atn, lowp)
                if lowp is None:
                    np.testing.assert_allclose(got, z[k + ".out"], rtol=0, atol=2e-6)
                else:
                    assert np.array_equal(got, z[k + ".out"]), (k, np.abs(got - z[k + ".out"]).max())
            for case in range(5):
                k = f"inject.{dt}.{case}"
                hard, soft, interp, _t, active = z[k + ".meta"]
                x = z[k + ".x"]
                got = O.inject_first_frame(x, 2, 16, None if hard else float(interp), lowp) if active else x
                if lowp is None:
                    np.testing.assert_allclose(got, z[k + ".out"], rtol=0, atol=1e-6)
                else:
                    assert np.array_equal(got, z[k + ".out"]), k
        finally:
            O.CPU_SCALAR_TENSOR_SEMANTICS = False


def test_clip_vision_oracle_matches_transformers_vectors(golden_dir):
    """oracle/clip_oracle.py clip_vision_forward against transformers' CLIPVisionModelWithProjection (tests/golden/clip_vision.npz)."""
    import torch
    from oracle import clip_oracle as CO
    z = np.load(os.path.join(golden_dir, "clip_vision.npz"))
    sd = {k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("sd.")}
    heads, patch = [int(v) for v in z["meta"]]
    o = CO.clip_vision_forward(sd, torch.from_numpy(z["pixel_values"]), heads, patch)
    torch.testing.assert_close(o["image_embeds"], torch.from_numpy(z["image_embeds"]), rtol=1e-4, atol=3e-5)
    # transformers returns the hidden state BEFORE post_layernorm as last_hidden_state
    torch.testing.assert_close(o["last_hidden_state"], torch.from_numpy(z["last_hidden_state"]), rtol=1e-4, atol=3e-5)
