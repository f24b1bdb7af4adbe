#!/usr/bin/env python3
"""Golden vectors for the segmentation side-car's mask post-processing (TEST INFRASTRUCTURE; build container only).

Runs the reference's text_segment/run_expand.py (script body, :25-87) unmodified via runpy with a stand-in
`lang_sam.LangSAM` whose predict() returns prepared boolean masks (GroundingDINO/SAM are not installed), and records
what the script saves for each concept.  Writes tests/golden/expand_masks.npz (inputs + outputs only)."""
import os
import runpy
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

REF = "/root/reference/text_segment/run_expand.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "expand_masks.npz")


def blob(H, W, cy, cx, ry, rx, rng):
    yy, xx = np.mgrid[0:H, 0:W]
    m = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
    m &= rng.rand(H, W) > 0.05                      # holes: the rectangle must come from the extent, not the area
    return m


def run_case(masks_in):
    saved = {}
    queue = list(masks_in)

    class FakeLangSAM:
        def predict(self, image_pil, prompt):
            m = queue.pop(0)
            return torch.from_numpy(m)[None], None, None, None

    mod = types.ModuleType("lang_sam")
    mod.LangSAM = FakeLangSAM
    sys.modules["lang_sam"] = mod
    tmp = tempfile.mkdtemp()
    H, W = masks_in[0].shape
    Image.fromarray(np.zeros((H, W, 3), np.uint8)).save(os.path.join(tmp, "in.png"))
    orig_save = Image.Image.save

    def rec_save(self, fp, *a, **k):
        if str(fp).endswith(".jpg"):
            saved[os.path.basename(str(fp))] = np.array(self).copy()
        return orig_save(self, fp, *a, **k)
    Image.Image.save = rec_save
    argv = sys.argv
    sys.argv = ["run_expand.py", f"--input_path={tmp}/in.png", "--text_condition=c0+c1", f"--output_path={tmp}"]
    try:
        runpy.run_path(REF, run_name="__main__")
    finally:
        sys.argv = argv
        Image.Image.save = orig_save
    return [saved["c0.jpg"], saved["c1.jpg"]]


def main():
    rng = np.random.RandomState(0)
    H = W = 96
    cases = {
        "disjoint": [blob(H, W, 30, 25, 14, 12, rng), blob(H, W, 60, 70, 18, 15, rng)],
        "overlap": [blob(H, W, 40, 35, 22, 20, rng), blob(H, W, 50, 55, 20, 22, rng)],
        "contained": [blob(H, W, 48, 48, 8, 8, rng), blob(H, W, 48, 48, 30, 30, rng)],     # >80 % of mask 0 in the overlap
        "touching": [blob(H, W, 30, 30, 10, 29, rng), blob(H, W, 62, 60, 21, 30, rng)],
    }
    out = {}
    for name, ms in cases.items():
        res = run_case([m.copy() for m in ms])
        for i in range(2):
            out[f"{name}_in{i}"] = ms[i]
            out[f"{name}_out{i}"] = res[i]
        print(name, [int(m.sum()) for m in ms], [int((r > 0).sum()) for r in res], res[0].dtype)
    np.savez_compressed(OUT, **out)


if __name__ == "__main__":
    main()
