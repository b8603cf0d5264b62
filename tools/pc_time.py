"""Tuning aid (GPU): wall time per iteration of the predictor-corrector loop (64 images, 64x64, bf16, reverse diffusion +
Langevin corrector) with one and with two launch chains: python tools/pc_time.py [n_corrector_steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd.modules import VESDE, PredictorCorrectorSampler, Unet  # noqa: E402

n_inner = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
u = Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
         compute_dtype="bf16").to(dev).eval()
N = 200
s = PredictorCorrectorSampler(predictor="reverse_diffusion", corrector="langevin", snr=0.16, n_steps=n_inner)
s.update_sde(VESDE(0.01, 50.0, N))
for chains in (1, 2, 1, 2):
    u.set_sample_chains(chains)
    ts = []
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = s.forward(u, [64, 3, 64, 64], dev, return_nfe=False, seed=5, keep="last", return_device_tensor=True)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / N * 1e6)
    print(f"PC_US chains={chains} n_corrector_steps={n_inner} per iteration: {' '.join(f'{t:.0f}' for t in ts)} us "
          f"({64 / (min(ts) * 1e-6):.0f} image-iterations/s, {N} iterations)", flush=True)
