"""Device time of one denoising step of the headline configuration (64x64, 64 images, bf16), from HIP events around a
graph-replayed chain: python tools/step_time.py [--steps 300] [--reps 3] [--lv] [--size 64] [--batch 64]"""
import argparse
import sys

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402
from diffusion_model_nemo_amd.modules import GaussianDiffusion, LearnedGaussianDiffusion, Unet  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--lv", action="store_true")
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--tag", default="")
ap.add_argument("--plan", type=int, default=None, help="planning batch of the launch heuristics")
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
u = Unet(input_dim=a.size, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
         learned_variance=a.lv, compute_dtype="bf16").to(dev).eval()
if a.plan is not None:
    u.set_plan_batch(a.plan)
gd = (LearnedGaussianDiffusion if a.lv else GaussianDiffusion)(timesteps=a.steps, schedule_name="cosine")
ms = []
host = []
for r in range(a.reps + 1):
    import time as _t
    t0 = _t.perf_counter()
    x = gd.p_sample_loop(u, (a.batch, 3, a.size, a.size), device=dev, seed=3, keep="last", return_device_tensor=True)
    host.append((_t.perf_counter() - t0) / a.steps * 1e6)
    torch.cuda.synchronize()
    ms.append(N.loop_last_ms(u.native_handle()) / a.steps * 1e3)
assert torch.isfinite(x).all()
print(f"STEP_US {a.tag} min {min(ms[1:]):.1f} all {' '.join(f'{m:.1f}' for m in ms[1:])}; host time of the call per step "
      f"{' '.join(f'{h:.0f}' for h in host[1:])} us", flush=True)
