"""GPU parity of the text towers (tweediemix_amd/text.py) against oracle/clip_oracle.py on the weights and ids of
tests/golden/clip_text.npz (which the oracle itself is pinned to, tests/test_oracle_golden.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(name, golden_dir):
    z = np.load(os.path.join(golden_dir, "clip_text.npz"))
    src = "g" if name == "e" else name
    sd = {k[len(src) + 4:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith(src + ".sd.")}
    heads, eos, gelu, _n = [int(v) for v in z[name + ".meta"]]
    return z, sd, heads, eos, ("gelu" if gelu else "quick_gelu")


def rel(a, b):
    return float((a.float().cpu() - b.float().cpu()).norm() / b.float().norm())


@pytest.mark.parametrize("name", ["l", "g", "e"])
def test_text_tower_matches_oracle(name, golden_dir):
    from tweediemix_amd import text as T
    from oracle import clip_oracle as CO
    z, sd, heads, eos, act = _case(name, golden_dir)
    ids = torch.from_numpy(z[name + ".ids"])
    enc = T.ClipTextEncoder(sd, heads, act, eos, device="cuda")
    hs, pooled = enc(ids)
    sd_bf = {k: (v.to(torch.bfloat16).float() if v.dim() == 2 and "embedding" not in k else v) for k, v in sd.items()}
    o = CO.clip_text_forward(sd_bf, ids, heads, act, eos_token_id=eos)
    want_pooled = o["text_embeds"] if o["text_embeds"] is not None else o["pooler_output"]
    assert rel(hs, o["hidden_states"][-2]) < 1e-2, rel(hs, o["hidden_states"][-2])
    assert rel(pooled, want_pooled) < 2e-2, rel(pooled, want_pooled)
    assert rel(hs, torch.from_numpy(z[name + ".hs_m2"])) < 2e-2            # and against the transformers vectors directly
    assert rel(enc.last_hidden_state(ids), torch.from_numpy(z[name + ".last"])) < 2e-2     # text_encoder(ids)[0] (video pipeline)
    assert rel(pooled, torch.from_numpy(z[name + ".pooled"])) < 3e-2


def test_encode_prompt_concat_and_token_injection(golden_dir):
    from tweediemix_amd import text as T
    from oracle import clip_oracle as CO
    zl, sdl, hl, el, al = _case("l", golden_dir)
    zg, sdg, hg, eg, ag = _case("g", golden_dir)
    ids_l, ids_g = torch.from_numpy(zl["l.ids"]), torch.from_numpy(zg["g.ids"])
    e1, e2 = T.ClipTextEncoder(sdl, hl, al, el), T.ClipTextEncoder(sdg, hg, ag, eg)
    # modifier-token injection (fusion_sampling.py:161-189): append a row, overwrite it, use its id
    new = torch.linspace(-1, 1, 128)
    for e, sd in ((e1, sdl), (e2, sdg)):
        n = e.tok.shape[0]
        e.resize_token_embeddings(n + 1)
        e.set_token_embedding(n, new)
        key = [k for k in sd if k.endswith("embeddings.token_embedding.weight")][0]
        sd[key] = torch.cat([sd[key], new[None]])
    ids_l, ids_g = ids_l.clone(), ids_g.clone()
    ids_l[0, 2] = ids_g[0, 2] = 64
    emb, pooled = T.encode_prompt([e1, e2], [ids_l, ids_g])
    want_e, want_p = CO.encode_prompt([(sdl, hl, al, el), (sdg, hg, ag, eg)], [ids_l, ids_g])
    assert emb.shape == (3, 77, 256) and pooled.shape == (3, 96)
    assert rel(emb, want_e) < 2e-2 and rel(pooled, want_p) < 3e-2


@pytest.mark.parametrize("tower", ["clip_l", "bigg"])
def test_full_size_towers_match_oracle(tower):
    """the two SDXL text towers at their real sizes (CLIP ViT-L/14: 12 x 768, 12 heads, quick_gelu; OpenCLIP bigG/14:
    32 x 1280, 20 heads, gelu, projection 1280) with random-init weights, 5 prompt rows of 77 tokens."""
    import time
    from tweediemix_amd import text as T, weights as Wt
    from oracle import clip_oracle as CO
    d, layers, inter, heads, act, proj = (768, 12, 3072, 12, "quick_gelu", None) if tower == "clip_l" else (1280, 32, 5120, 20, "gelu", 1280)
    sd = Wt.synthetic_clip_state_dict(d, layers, inter, vocab=49411, proj=proj, dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(1)
    ids = torch.full((5, 77), 49407 if tower == "clip_l" else 0, dtype=torch.long)
    for b, n in enumerate((3, 8, 14, 30, 75)):
        ids[b, 0] = 49406
        ids[b, 1:1 + n] = torch.randint(300, 49000, (n,), generator=g)
        ids[b, 1 + n] = 49407
    ids[1, 4] = 49409                                                   # an injected modifier token id (> EOS)
    enc = T.ClipTextEncoder(sd, heads, act, 2)
    hs, pooled = enc(ids)
    torch.cuda.synchronize()
    t0 = time.time()
    hs, pooled = enc(ids)
    torch.cuda.synchronize()
    print(f"{tower}: 5 prompts in {1e3 * (time.time() - t0):.1f} ms")
    o = CO.clip_text_forward({k: v.float() for k, v in sd.items()}, ids, heads, act, eos_token_id=2)
    want = o["text_embeds"] if proj else o["pooler_output"]
    assert rel(hs, o["hidden_states"][-2]) < 2e-2, rel(hs, o["hidden_states"][-2])
    assert rel(pooled, want) < 3e-2, rel(pooled, want)


def test_vision_tower_matches_oracle_and_transformers_vectors(golden_dir):
    """CLIP image tower with the ViT-H head size (80, zero-padded to 128 for the score GEMM) against oracle/clip_oracle.py and the
    transformers vectors of tests/golden/clip_vision.npz."""
    from tweediemix_amd import text as T
    from oracle import clip_oracle as CO
    z = np.load(os.path.join(golden_dir, "clip_vision.npz"))
    sd = {k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("sd.")}
    heads, patch = [int(v) for v in z["meta"]]
    x = torch.from_numpy(z["pixel_values"])
    enc = T.ClipVisionEncoder(sd, heads, patch)
    got = enc(x)
    sd_bf = {k: (v.to(torch.bfloat16).float() if v.dim() >= 2 and "embedding" not in k else v) for k, v in sd.items()}
    want = CO.clip_vision_forward(sd_bf, x, heads, patch)["image_embeds"]
    assert got.shape == (2, 64)
    assert rel(got, want) < 2e-2, rel(got, want)
    assert rel(got, torch.from_numpy(z["image_embeds"])) < 3e-2


def test_full_size_vision_tower_matches_oracle():
    """OpenCLIP ViT-H/14 image tower at its real size (32 x 1280, 16 heads of 80, 257 tokens, projection 1024), random-init weights."""
    from tweediemix_amd import text as T, weights as Wt
    from oracle import clip_oracle as CO
    sd = Wt.synthetic_clip_vision_state_dict(1280, 32, 5120, 16, dtype=torch.bfloat16)
    x = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(4))
    got = T.ClipVisionEncoder(sd, 16, 14)(x)
    want = CO.clip_vision_forward({k: v.float() for k, v in sd.items()}, x, 16, 14)["image_embeds"]
    assert got.shape == (1, 1024)
    assert rel(got, want) < 3e-2, rel(got, want)
