"""THE stated trajectory tolerance (north_star: "outputs match the reference latents within a stated fp tolerance"), at SDXL size.

The product sampler (hipGraph replay of the HIP launch plans, bf16 or e4m3 operands) against the oracle's sampler
(oracle/tweedie_oracle.py, pinned to the reference's own `denoise_step`) driving the fp32 UNet oracle on the same GPU:
SDXL-base parameter shapes, 512 x 512, n = 20, resampling 10, jumping 5 = BASELINE config 1's schedule (27 UNet calls at B = 4 +
18 at B = 2), K = 3 concepts.  Reference loop: fusion_sampling.py:490-494; call sites :342-374, 388-430.
The 1024 x 1024, n = 50 run of the same routine (75 calls) is recorded in profiles/ (tests/trajectory_parity.py --out).
"""
import pytest
import torch

from trajectory_parity import trajectory_parity

pytestmark = pytest.mark.gpu

# stated tolerance, rel L2 against the fp32 oracle trajectory on identical weights:
# measured (MI355X, round 5; 512^2 n = 20 here | 1024^2 n = 50 in profiles/r5_traj_1024_n50_*.json):
#   step    bf16 3.3e-3 | 3.8e-3      fp8 3.6e-3 | 4.4e-3
#   start   bf16 3.3e-2 | 3.3e-2      fp8 4.0e-2 | 4.0e-2
#   final   bf16 3.3e-2 | 4.3e-2      fp8 3.9e-2 | 6.3e-2
TOL_STEP = 1e-2            # one scheduler step started from the oracle's latent (one UNet call + the exact fused step)
TOL_START_STEP = 6e-2      # the start step: 1 + 2 * resampling_steps = 21 chained UNet calls at sqrt(alpha_t) ~ 0.07
TOL_FINAL = {"bf16": 6e-2, "fp8": 1e-1}      # final latent of the free-running trajectory (45 chained calls; 75 at 1024^2 n = 50)


@pytest.mark.parametrize("kind,fp8", [("lora", False), ("lora", True), ("custom", False)])
def test_full_size_trajectory_vs_oracle_sampler(sdxl_weights, kind, fp8):
    r = trajectory_parity(sdxl_weights, kind, res=512, n=20, fp8=fp8)
    per = r.pop("teacher_forced_per_step")
    print(f"full-size trajectory {kind} {'fp8' if fp8 else 'bf16'}: {r}")
    print("   teacher-forced per step: " + " ".join(f"{t}:{v:.2e}" for t, v in per))
    assert r["finite"] and r["same_unet_call_schedule"]
    # BASELINE config 1's schedule (27 @ B = 4 + 18 @ B = 2); the LoRA script's --t_stop 0.8 window hands the last 3 steps back to plain CFG calls
    assert (r["calls_BK1"], r["calls_B2"]) == ((24, 21) if kind == "lora" else (27, 18))
    assert r["teacher_forced_start_step"] <= TOL_START_STEP, r
    assert r["teacher_forced_worst_other_step"] <= TOL_STEP, r
    assert r["free_running_final_rel_l2"] <= TOL_FINAL[r["dtype"]], r
