"""Generates tests/golden/video_step.npz by EXECUTING the reference's own statements (read from /root/reference at
generation time only): the CFG + v-prediction DDIM update of the I2VGen-XL loop
(video_gen/pipeline_i2vgen_xl.py:480-482 `alpha`, :699-719) and the first-frame feature injection of the patched
ResnetBlock2D.forward (video_gen/utils_attn.py:433-455), on small seeded tensors, in fp32 and fp16.
The fixture holds inputs and outputs only.  Run in the build container: python oracle/gen_golden_video.py"""
import os, textwrap, types
import numpy as np
import torch
from einops import rearrange

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pl = open("/root/reference/video_gen/pipeline_i2vgen_xl.py").read().split("\n")
ua = open("/root/reference/video_gen/utils_attn.py").read().split("\n")
alpha_src = textwrap.dedent("\n".join(pl[479:482]))
step_src = textwrap.dedent("\n".join(pl[698:719]))
inject_src = textwrap.dedent("\n".join(ua[432:455]))
out = {}
g = torch.Generator().manual_seed(21)
# a stand-in alpha table (the real one comes from the checkpoint's scheduler config): cosine schedule with zero terminal SNR
N = 1000
ab = torch.cos((torch.arange(N + 1) / N + 0.008) / 1.008 * torch.pi / 2) ** 2
ac = (ab[1:] / ab[0]).clamp(min=0).float()
ac[-1] = 0.0
out["alphas_cumprod"] = ac.numpy()
for dt_name, dt in (("f32", torch.float32), ("f16", torch.float16)):
    for case, (t, skip, gs) in enumerate(((981, 20, 9.0), (501, 20, 9.0), (1, 20, 7.5), (21, 20, 1.5))):
        B, C, F, H, W = 1, 4, 16, 6, 5
        latents = torch.randn(B, C, F, H, W, generator=g).to(dt)
        noise_pred = torch.randn(2 * B, C, F, H, W, generator=g).to(dt)
        self = types.SimpleNamespace(scheduler=types.SimpleNamespace(alphas_cumprod=ac, step=lambda *a, **k: types.SimpleNamespace(pred_original_sample=None)),
                                     final_alpha_cumprod=ac[0].clone(), skip=skip, do_classifier_free_guidance=True)
        env = {"torch": torch}
        exec(alpha_src, env)
        self.alpha = types.MethodType(env["alpha"], self)
        env = {"self": self, "torch": torch, "noise_pred": noise_pred.clone(), "latents": latents.clone(), "t": torch.tensor(t),
               "guidance_scale": gs, "extra_step_kwargs": {}}
        exec(step_src, env)
        k = f"step.{dt_name}.{case}"
        out[k + ".x"], out[k + ".v"], out[k + ".out"] = latents.float().numpy(), noise_pred.float().numpy(), env["latents"].float().numpy()
        out[k + ".meta"] = np.array([t, skip, gs], np.float64)
    # first-frame injection: hard copy (mid_block resnets) and interpolation (up_blocks[1].resnets[0])
    for case, (sched, sched2, interp, t) in enumerate((([981, 961], None, None, 981), (None, [981, 961], 0.7, 961), ([981], None, None, 1000),
                                                       ([981], None, None, 941), (None, [981], 0.25, 1000))):
        x = torch.randn(32, 8, 3, 4, generator=g).to(dt)
        self = types.SimpleNamespace(injection_schedule=sched, injection_schedule2=sched2, interp=interp, t=t)
        env = {"self": self, "hidden_states": x, "output_tensor": x.clone(), "rearrange": rearrange, "print": lambda *a: None}
        exec(inject_src, env)
        k = f"inject.{dt_name}.{case}"
        out[k + ".x"], out[k + ".out"] = x.float().numpy(), env["output_tensor"].float().numpy()
        out[k + ".meta"] = np.array([1 if sched else 0, 1 if sched2 else 0, interp or 0.0, t, int(t in (sched or sched2)) or int(t == 1000)], np.float64)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "video_step.npz"), **out)
print("wrote", len(out), "arrays")

# ---- image side of the pipeline: _center_crop_wide / _resize_bilinear (:759-793) and prepare_image_latents (:421-451)
import PIL.Image
src = "\n".join(pl)
fn_src = src[src.index("def _resize_bilinear("):]
fn_src = fn_src[:fn_src.index("def _convert_pt_to_pil")] if "def _convert_pt_to_pil" in fn_src else fn_src
env = {"torch": torch, "PIL": PIL, "Union": __import__("typing").Union, "List": __import__("typing").List, "Tuple": __import__("typing").Tuple,
       "_convert_pt_to_pil": lambda im: im}
exec(fn_src, env)
rs = np.random.RandomState(5)
img = PIL.Image.fromarray(rs.randint(0, 256, (90, 150, 3), dtype=np.uint8))
out2 = {"img": np.array(img)}
for tag, res in (("sq", (64, 64)), ("wide", (96, 56))):
    out2[f"crop.{tag}"] = np.array(env["_center_crop_wide"](img, res))
out2["resize.224"] = np.array(env["_resize_bilinear"](env["_center_crop_wide"](img, (64, 64)), (32, 32)))
pil = textwrap.dedent("\n".join(pl[427:450]))          # body of prepare_image_latents after `image = image.to(device)`
class _Dist:
    def __init__(self, m): self.m = m
    def sample(self): return self.m
mean = torch.randn(1, 4, 7, 12, generator=g)
self = types.SimpleNamespace(vae=types.SimpleNamespace(encode=lambda im: types.SimpleNamespace(latent_dist=_Dist(mean)), config=types.SimpleNamespace(scaling_factor=0.18215)),
                             do_classifier_free_guidance=True)
env3 = {"self": self, "torch": torch, "image": torch.zeros(1), "device": "cpu", "num_frames": 16, "num_videos_per_prompt": 1}
exec(pil, env3)
out2["pil.mean"] = mean.numpy(); out2["pil.out"] = env3["image_latents"].numpy()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "video_image.npz"), **out2)
print("wrote video_image.npz", {k: v.shape for k, v in out2.items()})
