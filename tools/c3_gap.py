"""Where the improved-DDPM (learned variance) chain spends more time than the DDPM chain: both 64x64, 64 images, T=1000,
bf16, run as ten segments of 100 steps each; per segment the device time per step (HIP events on the loop's stream)
and the state's range. Usage: python tools/c3_gap.py"""
import sys

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402
from diffusion_model_nemo_amd.modules import GaussianDiffusion, LearnedGaussianDiffusion, Unet  # noqa: E402

dev = torch.device("cuda", 0)
B, S, T, SEG = 64, 64, 1000, 100
for lv in (False, True):
    torch.manual_seed(1234)
    u = Unet(input_dim=S, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
             learned_variance=lv, compute_dtype="bf16").to(dev).eval()
    gd = (LearnedGaussianDiffusion if lv else GaussianDiffusion)(timesteps=T, schedule_name="cosine")
    for rep in range(2):
        x = torch.empty((B, 3, S, S), device=dev)
        line = []
        for first in range(0, T, SEG):
            gd._run_chain(u, x, first, SEG, first == 0, "last", 5, 0)
            ms = N.loop_last_ms(u.native_handle()) / SEG
            fin = bool(torch.isfinite(x).all())
            line.append(f"{ms * 1e3:6.1f}us|{float(x.abs().max()) if fin else float('nan'):8.2e}")
        print(("improved" if lv else "ddpm    "), f"rep{rep}", " ".join(line), flush=True)
