"""Timing of the other baseline configurations on one GPU (bench.py measures only the headline configuration).

C3 shape: improved DDPM (learned variance), 64x64, 64 images per GPU, T=1000.
C4 shape: VE-SDE predictor-corrector, 32x32, B=64, N=1000: reverse_diffusion + langevin (2 NFE/step) and the YAML
default euler_maruyama + no corrector (1 NFE/step).
"""
import sys
import time

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd.modules import (GaussianDiffusion, LearnedGaussianDiffusion, PredictorCorrectorSampler,  # noqa: E402
                                              Unet, VESDE, GeneralizedGaussianDiffusion)

dev = torch.device("cuda", 0)


def net(side, lv=False, groups=8):
    torch.manual_seed(1234)
    return Unet(input_dim=side, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=groups, use_convnext=False,
                dropout=0.0, learned_variance=lv, compute_dtype="bf16").to(dev).eval()


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


B = 64
u = net(64, lv=True)
lgd = LearnedGaussianDiffusion(timesteps=1000, schedule_name="cosine")
el = timed(lambda: lgd.p_sample_loop(u, (B, 3, 64, 64), device=dev, seed=1, keep="last", return_device_tensor=True))
print(f"C3 improved DDPM 64x64 B=64 T=1000: {el:.3f} s/chain, {B / el:.1f} img/s per GPU", flush=True)

u = net(64)
ddim = GeneralizedGaussianDiffusion(timesteps=1000, schedule_name="cosine", eta=0.0, ddim_timesteps=10)
el = timed(lambda: ddim.p_sample_loop(u, (B, 3, 64, 64), device=dev, seed=1, keep="last", return_device_tensor=True))
print(f"DDIM 10 steps 64x64 B=64: {el * 1e3:.2f} ms/chain, {B / el:.0f} img/s", flush=True)

u = net(32, groups=4)
for pred, corr, nfe in (("reverse_diffusion", "langevin", 2), ("euler_maruyama", None, 1)):
    s = PredictorCorrectorSampler(predictor=pred, corrector=corr, snr=0.16, n_steps=1)
    s.update_sde(VESDE(0.01, 50.0, 1000))
    el = timed(lambda: s.sample(u, (B, 3, 32, 32), device=dev, keep="last", seed=1))
    print(f"C4 VE-SDE {pred}+{corr} 32x32 B=64 N=1000 ({nfe} NFE/step): {el:.3f} s/chain, {B / el:.1f} img/s", flush=True)
