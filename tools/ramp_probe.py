"""Experiment (GPU): is a denoising step slower at the start of a loop because the GPU was idle before it? Loops of 300
steps, each timed by its own HIP events: (a) with a host wait between loops, (b) enqueued back to back without one.
python tools/ramp_probe.py [one|two]"""
import sys

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402
from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(1234)
u = Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
         compute_dtype="bf16").to(dev).eval()
if len(sys.argv) > 1 and sys.argv[1] == "one":
    u.set_sample_chains(1)
steps = 300
gd = GaussianDiffusion(timesteps=steps, schedule_name="cosine")
shape = (64, 3, 64, 64)
gd.p_sample_loop(u, shape, device=dev, seed=3, keep="last", return_device_tensor=True)
torch.cuda.synchronize()
h = u.native_handle()
for mode in ("host wait between loops", "back to back", "host wait between loops", "back to back"):
    ms = []
    for rep in range(6):
        x = gd.p_sample_loop(u, shape, device=dev, seed=3, keep="last", return_device_tensor=True, check=False)
        if mode.startswith("host"):
            torch.cuda.synchronize()
            ms.append(N.loop_last_ms(h) / steps * 1e3)
    torch.cuda.synchronize()
    if not ms:
        ms.append(N.loop_last_ms(h) / steps * 1e3)   # the last of six loops that followed each other without a gap
    u.check_handoffs()
    print(f"RAMP {mode}: {' '.join(f'{m:.1f}' for m in ms)} us per step", flush=True)
