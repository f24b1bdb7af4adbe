import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from tweediemix_amd import vae as V
sd = V.synthetic_state_dict(V.FULL, device="cuda")
plan = V.VAEDecoderPlan(V.FULL, sd, 1, 128, 128, 1 / 0.13025)
z = torch.randn(1, 4, 128, 128, device="cuda") * 0.13
for _ in range(2): plan(z)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(5): img = plan(z)
torch.cuda.synchronize(); dt = (time.time() - t0) / 5
print(f"VAE decode 1024^2: {dt*1e3:.1f} ms, {plan.flops/1e12:.2f} TFLOP -> {plan.flops/dt/1e12:.0f} TFLOP/s; launches {len(plan.ops)}; finite {torch.isfinite(img).all().item()}")
