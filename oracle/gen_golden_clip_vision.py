"""Generates tests/golden/clip_vision.npz from transformers' CLIPVisionModelWithProjection (the image tower behind
`_encode_image` of the reference's video pipeline): tiny random-init config with the ViT-H head size (80), fp16-rounded weights.
Run in the build container: python oracle/gen_golden_clip_vision.py"""
import os
import numpy as np
import torch
from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
torch.manual_seed(0)
cfg = CLIPVisionConfig(hidden_size=320, intermediate_size=320, num_hidden_layers=1, num_attention_heads=4, image_size=56, patch_size=14,
                       hidden_act="gelu", projection_dim=64)
m = CLIPVisionModelWithProjection(cfg).eval()
with torch.no_grad():
    for p in m.parameters():
        p.copy_((p * (3.0 if p.dim() > 1 else 1.0)).half().float())
x = torch.randn(2, 3, 56, 56, generator=torch.Generator().manual_seed(2))
with torch.no_grad():
    o = m(pixel_values=x)
out = {"pixel_values": x.numpy(), "image_embeds": o.image_embeds.numpy(), "last_hidden_state": o.last_hidden_state.numpy(),
       "meta": np.array([4, 14])}
for k, v in m.state_dict().items():
    out["sd." + k] = v.half().numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "clip_vision.npz"), **out)
print("wrote", sum(v.nbytes for v in out.values()) / 1e6, "MB")
