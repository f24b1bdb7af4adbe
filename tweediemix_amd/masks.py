"""Blend-mask preparation (host side, once per image).

preprocess_mask / build_masks follow fusion_generation/fusion_sampling.py:81-89 and :461-469: the
segmentation side-car's '<concept>.jpg' (8-bit grey) -> /255 -> threshold 0.5 -> nearest resize to the
latent grid -> masks = [fg_1..fg_{K-1}, clamp(1 - sum fg, 0)].  random_rectangle_masks is the synthetic
stand-in for the side-car used by the benchmark (run_expand.py:50-51 emits bounding rectangles).
"""
from __future__ import annotations

import numpy as np
import torch


def preprocess_mask(mask, h: int, w: int, device="cuda") -> torch.Tensor:
    """mask: path or uint8 array [H,W]. Returns [1,1,h,w] fp32 {0,1} on `device`."""
    if isinstance(mask, (str, bytes)):
        from PIL import Image
        mask = np.array(Image.open(mask).convert("L"))
    m = np.asarray(mask).astype(np.float32) / np.float32(255.0)
    m = (m >= 0.5).astype(np.float32)
    H, W = m.shape
    ys = np.minimum(np.floor(np.arange(h, dtype=np.float32) * np.float32(H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w, dtype=np.float32) * np.float32(W / w)).astype(np.int64), W - 1)
    return torch.from_numpy(np.ascontiguousarray(m[ys][:, xs]))[None, None].to(device)


def build_masks(fg_sources, h: int, w: int, device="cuda") -> torch.Tensor:
    """[K,1,h,w]: foreground masks then background = clamp(1 - sum fg, min 0) (NOT renormalised)."""
    fg = torch.cat([preprocess_mask(s, h, w, device) for s in fg_sources])
    bg = 1 - torch.sum(fg, dim=0, keepdim=True)
    bg[bg < 0] = 0
    return torch.cat([fg, bg]).contiguous()


def random_rectangle_masks(K: int, H: int, W: int, seed: int = 0):
    """K-1 seeded axis-aligned rectangles (10-30 % area each) on the H x W image grid, uint8 {0,255}."""
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(K - 1):
        area = rng.uniform(0.10, 0.30) * H * W
        ar = rng.uniform(0.6, 1.6)
        hh = int(min(H, max(8, round((area * ar) ** 0.5))))
        ww = int(min(W, max(8, round(area / hh))))
        y0 = rng.randint(0, H - hh + 1)
        x0 = rng.randint(0, W - ww + 1)
        m = np.zeros((H, W), np.uint8)
        m[y0:y0 + hh, x0:x0 + ww] = 255
        out.append(m)
    return out


def partition_rectangle_masks(K: int, H: int, W: int, seed: int = 0):
    """K-1 seeded rectangles that do NOT overlap: rectangle k loses every pixel an earlier one claimed (what the side-car's
    overlap rule does to its second mask above the 0.8 threshold, text_segment/run_expand.py:78-87), re-drawn while less than
    half of it survives.  With these, fg_1 + ... + fg_{K-1} + bg == 1 on every pixel, i.e. the blend of
    fusion_sampling.py:466-469 is a convex combination and the latent keeps its scale through the fusion window; with
    random_rectangle_masks the weights sum to 2 on the overlap and that region doubles every fusion step (the reference does
    not normalise).  uint8 {0,255}, same conventions as random_rectangle_masks."""
    rng = np.random.RandomState(seed)
    taken = np.zeros((H, W), bool)
    out = []
    for _ in range(K - 1):
        for _try in range(64):
            area = rng.uniform(0.10, 0.30) * H * W
            ar = rng.uniform(0.6, 1.6)
            hh = int(min(H, max(8, round((area * ar) ** 0.5))))
            ww = int(min(W, max(8, round(area / hh))))
            y0 = rng.randint(0, H - hh + 1)
            x0 = rng.randint(0, W - ww + 1)
            m = np.zeros((H, W), bool)
            m[y0:y0 + hh, x0:x0 + ww] = True
            m &= ~taken
            if 2 * int(m.sum()) >= hh * ww:
                break
        taken |= m
        out.append(m.astype(np.uint8) * 255)
    return out


def synthetic_masks(kind: str, K: int, H: int, W: int, seed: int = 0):
    """'partition' (default of bench.py and the trajectory parity test) or 'overlap' (random_rectangle_masks: the reference's
    un-normalised weights on intersecting rectangles, kept as a labelled second case)."""
    if kind == "partition":
        return partition_rectangle_masks(K, H, W, seed)
    if kind == "overlap":
        return random_rectangle_masks(K, H, W, seed)
    raise ValueError(f"mask kind {kind!r}: 'partition' or 'overlap'")


def expand_masks(masks):
    """The side-car's post-processing of the two SAM masks (text_segment/run_expand.py:35-87): every mask becomes its
    bounding rectangle; where the two rectangles overlap, the bounding box of the overlap is re-filled with the
    ORIGINAL masks restricted to the overlap, and mask 1 loses it entirely when more than 80 % of original mask 0
    lies inside the overlap.  Like the reference this handles exactly two foreground masks (:62); for any other
    count only the rectangles are produced.  masks: list of bool arrays [H,W]; returns list of bool arrays."""
    import numpy as np
    orig = [np.asarray(m).astype(bool) for m in masks]
    rect = []
    for m in orig:
        ys, xs = np.nonzero(m)
        r = np.zeros_like(m)
        r[ys.min():ys.max() + 1, xs.min():xs.max() + 1] = True
        rect.append(r)
    if len(rect) == 2:
        ov = rect[0] & rect[1]
        if ov.any():
            ys, xs = np.nonzero(ov)
            y0, y1, x0, x1 = ys.min(), ys.max() + 1, xs.min(), xs.max() + 1
            o1 = ov & orig[0]
            o2 = ov & orig[1]
            if o1.sum() / orig[0].sum() > 0.8:
                o2 = np.zeros_like(o2)
            rect[0][y0:y1, x0:x1] = o1[y0:y1, x0:x1]
            rect[1][y0:y1, x0:x1] = o2[y0:y1, x0:x1]
    return rect


def sidecar_layout(output_path, rank: int, world: int, local_gpu: int, seg_gpu: int):
    """(directory, GPU) the segmentation side-car of rank `rank` uses.  One process (the reference's mode): exactly the reference's
    `{output_path}` and `--seg_gpu`.  Several ranks sample different seeds at the same time, so each gets its OWN directory
    `{output_path}/rank{r}` -- with a shared one, rank A could read the masks the side-car wrote for rank B's preview -- and, when
    `--seg_gpu` names a GPU that one of the ranks samples on (the default 1 does as soon as world > 1), the side-car runs on the
    rank's own GPU instead: that GPU is idle while its rank waits for the masks, whereas another rank's GPU is mid-trajectory."""
    import os
    if world <= 1:
        return output_path, seg_gpu
    gpu = local_gpu if 0 <= seg_gpu < world else seg_gpu
    # the side-car's command line sets CUDA_VISIBLE_DEVICES itself (fusion_sampling.py:458), and the child inherits the parent's environment:
    #   * parent restricted by CUDA_VISIBLE_DEVICES: the child's own value replaces it -> the i-th entry of the parent's list;
    #   * parent restricted by HIP_VISIBLE_DEVICES: the HIP runtime prefers that variable, so a child that inherits it would ignore its own
    #     CUDA_VISIBLE_DEVICES and every side-car would land on the first visible GPU -> the i-th entry of the HIP list, and
    #     SidecarMaskProvider drops HIP_VISIBLE_DEVICES from the child's environment (sidecar_child_env);
    #   * parent restricted by ROCR_VISIBLE_DEVICES only: HIP / CUDA ordinals index the already filtered list -> the rank-local index as is.
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")
    if vis and gpu == local_gpu:
        ids = [v.strip() for v in vis.split(",") if v.strip()]
        if 0 <= local_gpu < len(ids) and ids[local_gpu].lstrip("-").isdigit():
            gpu = int(ids[local_gpu])
    return os.path.join(output_path, f"rank{rank}"), gpu


def sidecar_child_env():
    """environment of the side-car process: the parent's without HIP_VISIBLE_DEVICES (see sidecar_layout: the command line's own
    CUDA_VISIBLE_DEVICES must be what selects the GPU; ROCR_VISIBLE_DEVICES, which filters below both, stays)"""
    import os
    env = dict(os.environ)
    env.pop("HIP_VISIBLE_DEVICES", None)
    return env


class SidecarMaskProvider:
    """The reference's segmentation side-car contract (fusion_sampling.py:453-469): decode the Tweedie preview,
    save `{output_path}/tweedie.jpg`, run an external command that writes `{output_path}/{seg_concept}.jpg`
    (8-bit masks; `text_segment/run_expand.py` in the reference), read them back through preprocess_mask.
    cmd_template may use {input_path}, {text_condition}, {output_path}, {seg_gpu}."""

    DEFAULT_CMD = ('CUDA_VISIBLE_DEVICES={seg_gpu} python text_segment/run_expand.py --input_path={input_path} '
                   '--text_condition="{text_condition}" --output_path={output_path}')

    def __init__(self, sampler, output_path, seg_concepts, seg_gpu=1, cmd_template=None):
        self.sampler, self.output_path, self.seg_concepts, self.seg_gpu = sampler, output_path, seg_concepts, seg_gpu
        self.cmd_template = cmd_template or self.DEFAULT_CMD

    def __call__(self, x0_preview):
        import os
        from PIL import Image
        os.makedirs(self.output_path, exist_ok=True)
        assert x0_preview.shape[0] == 1, "one preview per call: the sampler asks once per co-batched seed, in seed order"
        img = self.sampler.decode_latent(x0_preview[:1])[0]                       # [3,H,W] in [0,1]
        arr = (img.clamp(0, 1) * 255).to(torch.uint8).permute(1, 2, 0).cpu().numpy()   # ToPILImage semantics (mul 255, byte)
        path = os.path.join(self.output_path, "tweedie.jpg")
        Image.fromarray(arr).save(path)
        cmd = self.cmd_template.format(input_path=path, text_condition=self.seg_concepts, output_path=self.output_path,
                                       seg_gpu=self.seg_gpu)
        import subprocess
        subprocess.call(cmd, shell=True, env=sidecar_child_env())                # (os.system with a cleaned environment; return code ignored, like the reference)
        paths = [os.path.join(self.output_path, sp + ".jpg") for sp in self.seg_concepts.split("+")]
        s = self.sampler
        return build_masks(paths, s.h, s.w, s.device)
