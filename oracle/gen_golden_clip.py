"""Generates tests/golden/clip_text.npz from the transformers package (the reference's third-party dependency for
`encode_prompt`, fusion_sampling.py:43-68): tiny random-init CLIPTextModel / CLIPTextModelWithProjection, fp16-rounded
weights, fixed token ids (some rows carry ids above the EOS id, like added modifier tokens).  Run in the build container:
    python oracle/gen_golden_clip.py"""
import os, sys
import numpy as np
import torch
from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
torch.manual_seed(0)
V, S = 64, 77
specs = {"l": dict(cls=CLIPTextModel, act="quick_gelu", eos=2, proj=None),
         "g": dict(cls=CLIPTextModelWithProjection, act="gelu", eos=2, proj=96),
         "e": dict(cls=CLIPTextModelWithProjection, act="gelu", eos=V - 4, proj=96)}
for name, sp in specs.items():
    cfg = CLIPTextConfig(vocab_size=V, hidden_size=128, intermediate_size=128, num_hidden_layers=2, num_attention_heads=2,
                         max_position_embeddings=S, hidden_act=sp["act"], eos_token_id=sp["eos"], bos_token_id=V - 5, pad_token_id=1,
                         projection_dim=sp["proj"] or 128)
    m = sp["cls"](cfg).eval()
    if name == "e":                                                            # same weights as "g", other pooled rule
        m.load_state_dict(g_state)
    else:
        with torch.no_grad():
            for p in m.parameters():
                p.copy_((p * (3.0 if p.dim() > 1 else 1.0)).half().float())  # larger weights: attention far from uniform
    if name == "g":
        g_state = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    eos = V - 4
    ids = torch.full((3, S), eos if name != "g" else 0, dtype=torch.long)      # pad = eos (CLIP-L) / 0 ('!', bigG)
    for b, n in enumerate((5, 9, 20)):
        ids[b, 0] = V - 5
        ids[b, 1:1 + n] = torch.randint(3, V - 6, (n,), generator=g)
        ids[b, 1 + n] = eos
    ids[1, 3] = V - 2                                                          # an added modifier token (id > eos)
    ids[2, 7] = V - 1
    with torch.no_grad():
        o = m(input_ids=ids, output_hidden_states=True)
    pooled = o.text_embeds if sp["proj"] else o.pooler_output
    out[f"{name}.ids"] = ids.numpy()
    out[f"{name}.hs_m2"] = o.hidden_states[-2].numpy()
    out[f"{name}.last"] = o.last_hidden_state.numpy()
    out[f"{name}.pooled"] = pooled.numpy()
    out[f"{name}.meta"] = np.array([2, sp["eos"], 1 if sp["act"] == "gelu" else 0, len(o.hidden_states)])
    if name != "e":
        for k, v in m.state_dict().items():
            out[f"{name}.sd.{k}"] = v.half().numpy()
    out[f"{name}.hs_m2"] = out[f"{name}.hs_m2"].astype(np.float32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clip_text.npz"), **out)
print("wrote", sum(v.nbytes for v in out.values()) / 1e6, "MB raw")
