"""THE stated trajectory tolerance (north_star: "outputs match the reference latents within a stated fp tolerance"), at SDXL size.

The product sampler (hipGraph replay of the HIP launch plans, bf16 or e4m3 operands) against the oracle's sampler
(oracle/tweedie_oracle.py, pinned to the reference's own `denoise_step`) driving the fp32 UNet oracle on the same GPU:
SDXL-base parameter shapes, 512 x 512, n = 20, resampling 10, jumping 5 = BASELINE config 1's schedule (27 UNet calls at B = 4 +
18 at B = 2), K = 3 concepts.  Reference loop: fusion_sampling.py:490-494; call sites :342-374, 388-430; blend :466-469.
The 1024 x 1024, n = 50 run of the same routine (75 calls) is recorded in profiles/ (tests/trajectory_parity.py --out).

Round 6: the default masks PARTITION the image (blend weights sum to 1, the latent keeps its scale through the fusion window, so the
window's 13 - 16 B = 4 calls are visible in every number below); the overlapping rectangles of round 5 (weights sum to 2 on the
overlap: the reference does not normalise, the latent doubles there every fusion step) are a second, labelled case.
"""
import pytest
import torch

from trajectory_parity import trajectory_parity

pytestmark = pytest.mark.gpu

# stated tolerance, against the fp32 oracle trajectory on identical weights, partition masks.
# measured (MI355X, round 6, 512^2 n = 20; bf16 LoRA | fp8 LoRA | bf16 Custom-Diffusion):
#   eps per call, oracle's latent      3.6e-3 (4.5e-3 at the start step) | 3.6e-3 (5.0e-3) | 3.6e-3 (4.5e-3)     -- the same at every one of the 20 steps
#   x' per step, oracle's latent       3.3e-3 falling to 7e-5 (the update shrinks from 21 % to 1.2 % of x') | 3.6e-3 | 3.3e-3;   start step 3.3e-2 | 4.0e-2 | 3.3e-2
#   error / update per step            1.6e-2 ... 2.2e-2 | 1.7e-2 ... 2.2e-2 | 1.6e-2 ... 2.2e-2                   -- fusion window included (2.2e-2 is inside it)
#   eps per call, free-running         3.2e-2 ... 3.7e-2 | 3.9e-2 ... 4.4e-2 | 3.2e-2 ... 3.7e-2                   -- the start step's drift, carried
#   final latent, free-running         3.8e-2 | 4.6e-2 | 3.8e-2;   max|x| along the trajectory 38.6 (x_T: 4.2)
TOL_EPS_CALL = 2e-2        # eps of one UNet call on the oracle's own latent (the per-call bound of every whole-UNet test), every scheduler step
TOL_STEP = 1e-2            # rel L2 of x' after one scheduler step started from the oracle's latent (one UNet call + the exact fused step)
TOL_STEP_UPDATE = 4e-2     # the same error over the step's UPDATE ||x' - sqrt(a'/a) x|| (the part of x' that eps produced: -a e_cfg + b e_uncond with a ~ b, so the
                           # update is a difference of nearly equal terms and a 3.6e-3 error of eps shows as 1.6e-2 ... 2.2e-2 of it)
TOL_START_STEP = 6e-2      # the start step: 1 + 2 * resampling_steps = 21 chained UNet calls at sqrt(alpha_t) ~ 0.07
TOL_FINAL = {"bf16": 6e-2, "fp8": 1e-1}      # final latent of the free-running trajectory (45 chained calls; 75 at 1024^2 n = 50)
TOL_EPS_FREE_FUSION = {"bf16": 8e-2, "fp8": 1.2e-1}   # eps of the fusion window's calls along the two FREE-RUNNING trajectories (carries the latent drift)
FLOOR_VISIBLE = 1e-5       # a teacher-forced error below this inside the fusion window (t > 1) means the metric no longer sees eps (round 5's degenerate case: 6e-15)
FLOOR_UPDATE_FRACTION = 1e-3   # ... and so does an update that is less than this fraction of ||x'||
MAX_ABS_LATENT = 60.0      # |x| along the oracle's trajectory with partition masks (x_T ~ N(0,1); the final latent is an x0 estimate)


@pytest.mark.parametrize("kind,fp8", [("lora", False), ("lora", True), ("custom", False)])
def test_full_size_trajectory_vs_oracle_sampler(sdxl_weights, sdxl_bundles, kind, fp8):
    r = trajectory_parity(sdxl_weights, kind, res=512, n=20, fp8=fp8, mask_kind="partition", bundle=sdxl_bundles(kind))
    per = r.pop("teacher_forced_per_step")
    upd = r.pop("teacher_forced_per_step_update_normalised")
    eps1 = r.pop("teacher_forced_eps_first_call_per_step")
    calls = r.pop("free_running_eps_per_call")
    mx = r.pop("max_abs_latent_oracle_per_step")
    print(f"full-size trajectory {kind} {'fp8' if fp8 else 'bf16'} (partition masks): {r}")
    print("   teacher-forced x' rel L2 per step:      " + " ".join(f"{t}:{v:.2e}" for t, v in per))
    print("   teacher-forced error / update per step: " + " ".join(f"{t}:{v:.2e}" for t, v, _u in upd))
    print("   update / ||x'|| per step:               " + " ".join(f"{t}:{u:.2e}" for t, _v, u in upd))
    print("   teacher-forced eps (first call) per step:" + " ".join(f"{t}:{v:.2e}" for t, v in eps1))
    print("   free-running eps per call (B,t,phase):  " + " ".join(f"{B}@{t}{p[0]}:{v:.2e}" for _i, B, t, p, v in calls))
    print("   max|x| along the oracle's trajectory:   " + " ".join(f"{v:.3g}" for v in mx))
    assert r["finite"] and r["same_unet_call_schedule"]
    # BASELINE config 1's schedule (27 @ B = 4 + 18 @ B = 2); the LoRA script's --t_stop 0.8 window hands the last 3 steps back to plain CFG calls
    assert (r["calls_BK1"], r["calls_B2"]) == ((24, 21) if kind == "lora" else (27, 18))
    assert r["mask_weight_sum_min_max"] == [1.0, 1.0] and r["mask_overlap_fraction"] == 0.0
    assert r["max_abs_latent_oracle"] <= MAX_ABS_LATENT and r["max_abs_latent_product_final"] <= MAX_ABS_LATENT, r
    assert r["teacher_forced_worst_eps_first_call"] <= TOL_EPS_CALL, r
    assert r["teacher_forced_start_step"] <= TOL_START_STEP, r
    assert r["teacher_forced_worst_other_step"] <= TOL_STEP, r
    assert r["teacher_forced_worst_other_step_update_normalised"] <= TOL_STEP_UPDATE, r
    fw = r["fusion_window"]
    assert fw["steps"] == (13 if kind == "lora" else 16)      # the LoRA window [t_cond, t_stop] keeps its off-by-one step at t_stop (fusion_base plan)
    assert fw["teacher_forced_min"] >= FLOOR_VISIBLE and fw["update_fraction_min"] >= FLOOR_UPDATE_FRACTION, f"the fusion window is invisible to the step metric: {fw}"
    assert fw["update_normalised_max"] <= TOL_STEP_UPDATE and fw["eps_first_call_max"] <= TOL_EPS_CALL, fw
    assert r["free_running_eps_worst_fusion"] <= TOL_EPS_FREE_FUSION[r["dtype"]], r
    assert r["free_running_final_rel_l2"] <= TOL_FINAL[r["dtype"]], r


def test_full_size_trajectory_overlapping_masks_reference_quirk(sdxl_weights, sdxl_bundles):
    """LABELLED second case: intersecting rectangles, un-normalised weights (fusion_sampling.py:466-469 sums to 2 on the overlap): the oracle's
    latent grows by 2x per fusion step there and the product has to follow it -- same call schedule, finite, final latent within the bf16 bound.
    (This is round 5's only case; its per-step numbers say little about the fusion window, which is why it is no longer the stated tolerance.)"""
    r = trajectory_parity(sdxl_weights, "lora", res=512, n=20, fp8=False, mask_kind="overlap", bundle=sdxl_bundles("lora"), teacher_forced=False)
    r.pop("free_running_eps_per_call")
    mx = r.pop("max_abs_latent_oracle_per_step")
    print(f"full-size trajectory lora bf16 (OVERLAPPING masks, weights sum to 2 on {100 * r['mask_overlap_fraction']:.1f} % of the latent): {r}")
    print("   max|x| along the oracle's trajectory:   " + " ".join(f"{v:.3g}" for v in mx))
    assert r["finite"] and r["same_unet_call_schedule"]
    assert r["mask_weight_sum_min_max"][1] == 2.0 and r["mask_overlap_fraction"] > 0.02
    assert r["max_abs_latent_oracle"] > 100.0          # the explosion is the reference's arithmetic, reproduced
    assert r["free_running_final_rel_l2"] <= TOL_FINAL["bf16"], r
