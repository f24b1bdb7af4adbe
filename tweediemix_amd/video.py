"""Host side of the video sampler's per-step path (BASELINE config #5): the denoising loop of
video_gen/pipeline_i2vgen_xl.py:647-719 and the feature-injection schedule of video_gen/utils_attn.py:14-23,389-474, over
the HIP kernels `tmix_vpred_step` and `tmix_frame_inject`.  The I2VGen-XL UNet itself (diffusers `I2VGenXLUNet`, not in
/root/reference) is built natively in tweediemix_amd/i2vgen.py (I2VPlan / I2VPlanGroup); the loop here takes the network as a
callable, so that plan (as run_video.py wires it) or any other module plugs in.

Quirks kept: alpha(t) indexes the UN-shifted alphas_cumprod (unlike the image sampler) and falls back to
final_alpha_cumprod below 0 (:480-482); skip = 1000 // n (:647); the injection schedule is the first int(n * ratio)
timesteps, also active when t == 1000 (utils_attn.py:433,444); clips are hard-wired to b=2, t=16 there (:439,449)."""
from __future__ import annotations

import numpy as np
import torch

from . import ops


class VideoSchedule:
    """alphas_cumprod [1000] (from the checkpoint's scheduler), leading timesteps with steps_offset (diffusers DDIMScheduler
    set_timesteps as configured for i2vgen-xl), skip = 1000 // n."""

    def __init__(self, alphas_cumprod, n_steps: int, steps_offset: int = 1, set_alpha_to_one: bool = False):
        self.acp = np.asarray(alphas_cumprod, np.float32)
        self.final_alpha_cumprod = np.float32(1.0) if set_alpha_to_one else self.acp[0]
        self.n = n_steps
        self.skip = len(self.acp) // n_steps
        self.timesteps = (np.arange(0, n_steps) * self.skip).round()[::-1].astype(np.int64) + steps_offset

    def alpha(self, t: int):
        return self.acp[int(t)] if t >= 0 else self.final_alpha_cumprod

    def injection_schedule(self, ratio: float):
        k = int(self.n * ratio)
        return set(int(t) for t in self.timesteps[:k]) if k >= 0 else set()


def alphas_from_scheduler_config(cfg: dict):
    """alphas_cumprod (fp32 [num_train_timesteps]) and the VideoSchedule keyword arguments a diffusers DDIMScheduler built
    from `scheduler/scheduler_config.json` would hold (what `pipe.scheduler.alphas_cumprod` is at pipeline_i2vgen_xl.py:480):
    the beta schedule, `rescale_betas_zero_snr`, `steps_offset` and `set_alpha_to_one` follow the checkpoint, in the
    float32 arithmetic diffusers uses (torch.linspace / cumprod / sqrt on float32 tensors)."""
    import math
    T = int(cfg.get("num_train_timesteps", 1000))
    b0, b1 = float(cfg.get("beta_start", 0.0001)), float(cfg.get("beta_end", 0.02))
    sched = cfg.get("beta_schedule", "linear")
    if cfg.get("trained_betas") is not None:
        betas = torch.tensor(cfg["trained_betas"], dtype=torch.float32)
    elif sched == "linear":
        betas = torch.linspace(b0, b1, T, dtype=torch.float32)
    elif sched == "scaled_linear":
        betas = torch.linspace(b0 ** 0.5, b1 ** 0.5, T, dtype=torch.float32) ** 2
    elif sched == "squaredcos_cap_v2":                       # betas_for_alpha_bar: python floats, then one float32 tensor
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        betas = torch.tensor([min(1 - ab((i + 1) / T) / ab(i / T), 0.999) for i in range(T)], dtype=torch.float32)
    else:
        raise ValueError(f"beta_schedule {sched!r} is not one of diffusers DDIMScheduler's")
    if cfg.get("rescale_betas_zero_snr", False):             # rescale_zero_terminal_snr
        abs_ = torch.cumprod(1.0 - betas, dim=0).sqrt()
        a0, aT = abs_[0].clone(), abs_[-1].clone()
        abs_ = (abs_ - aT) * (a0 / (a0 - aT))
        bar = abs_ ** 2
        alphas = torch.cat([bar[0:1], bar[1:] / bar[:-1]])
        betas = 1 - alphas
    acp = torch.cumprod(1.0 - betas, dim=0).numpy().astype(np.float32)
    return acp, dict(steps_offset=int(cfg.get("steps_offset", 0)), set_alpha_to_one=bool(cfg.get("set_alpha_to_one", True)))


def injection_active(t: int, schedule) -> bool:
    return schedule is not None and (int(t) in schedule or int(t) == 1000)


class FeatureInjector:
    """what `register_conv_control_efficient` + `register_time` do to mid_block.resnets[0,1] (hard copy of the first
    frame's features) and up_blocks[1].resnets[0] (interp blend): call `apply(site, features)` on the resnet OUTPUT
    [(2*16), ...] of those three modules (e.g. from a forward hook)."""

    SITES = {"mid_block.resnets.0": "hard", "mid_block.resnets.1": "hard", "up_blocks.1.resnets.0": "interp"}

    def __init__(self, schedule, interp: float, clips: int = 2, frames: int = 16):
        self.schedule, self.interp, self.clips, self.frames = schedule, interp, clips, frames
        self.t = None

    def register_time(self, t: int):
        self.t = int(t)

    def apply(self, site: str, features: torch.Tensor) -> torch.Tensor:
        if injection_active(self.t, self.schedule):
            ops.frame_inject(features, self.clips, self.frames, None if self.SITES[site] == "hard" else self.interp)
        return features


@torch.no_grad()
def sample_loop(unet, latents: torch.Tensor, schedule: VideoSchedule, guidance_scale: float, injector: FeatureInjector | None = None):
    """pipeline_i2vgen_xl.py:680-719.  unet(latent_model_input [2B,C,F,H,W], t) -> v-prediction of the same shape (the
    caller closes over prompt/image conditioning); latents [B,C,F,H,W] on the GPU in the model dtype."""
    x = latents.contiguous()
    for t in schedule.timesteps:
        if injector is not None:
            injector.register_time(t)
            plan = getattr(unet, "plan", None)             # a native I2VPlan carries the injection as ops of its forward
            if plan is not None:
                plan.inject, plan.interp = injection_active(int(t), injector.schedule), injector.interp
        v = unet(torch.cat([x, x]), int(t)).contiguous()
        x = ops.vpred_step(x, v, guidance_scale, schedule.alpha(int(t)), schedule.alpha(int(t) - schedule.skip))
    return x


# ------------------------------------------------------------------------------------------ image side of the pipeline (host)
def center_crop_wide(image, resolution):
    """video_gen/pipeline_i2vgen_xl.py:772-793: BOX-resize so the image covers `resolution` (w, h), then centre crop (PIL)."""
    import PIL.Image
    scale = min(image.size[0] / resolution[0], image.size[1] / resolution[1])
    image = image.resize((round(image.width // scale), round(image.height // scale)), resample=PIL.Image.BOX)
    x1 = (image.width - resolution[0]) // 2
    y1 = (image.height - resolution[1]) // 2
    return image.crop((x1, y1, x1 + resolution[0], y1 + resolution[1]))


def resize_bilinear(image, resolution):
    """:759-769."""
    import PIL.Image
    return image.resize(resolution, PIL.Image.BILINEAR)


def clip_pixel_values(image, mean=(0.48145466, 0.4578275, 0.40821073), std=(0.26862954, 0.26130258, 0.27577711)):
    """`_encode_image` (:300-315): PIL -> [0,1] float -> CLIP normalisation, no further resize / crop.  [1,3,H,W] fp32."""
    a = torch.from_numpy(np.asarray(image, np.float32) / 255.0).permute(2, 0, 1)[None]
    return (a - torch.tensor(mean)[None, :, None, None]) / torch.tensor(std)[None, :, None, None]


def vae_pixel_values(image):
    """VideoProcessor.preprocess: PIL -> [-1,1] float, [1,3,H,W]."""
    return torch.from_numpy(np.asarray(image, np.float32) / 255.0).permute(2, 0, 1)[None] * 2.0 - 1.0


def prepare_image_latents(latent_sample, num_frames, scaling_factor=0.18215, cfg=True):
    """:421-451 after `vae.encode(image).latent_dist.sample()`: scale, add the frame axis, append one constant plane per later
    frame holding its position (frame_idx+1)/(num_frames-1) in all 4 channels, duplicate for classifier-free guidance."""
    il = (latent_sample * scaling_factor).unsqueeze(2)
    masks = [torch.ones_like(il[:, :, :1]) * ((f + 1) / (num_frames - 1)) for f in range(num_frames - 1)]
    if masks:
        il = torch.cat([il, torch.cat(masks, dim=2)], dim=2)
    return torch.cat([il] * 2) if cfg else il
