"""Tuning aid (GPU): phase times of ONE convolution launch measured where it runs -- inside the graph-replayed denoising
steps, both launch chains in flight -- from the per-workgroup stamps of the kernel (set-up / staging / matrix loop /
epilogue, in cycles; life and launch span on the 100 MHz wall clock).

usage: python tools/insitu_phases.py <site> [<site> ...]   (site = number of the launch among an evaluation's convolution
launches: 0 = downs.0 block1.proj, 1 = its block2.proj, ...; the process is re-run per site: the choice is read when the
handle is created)"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np


def one(site):
    os.environ["DMN_PROBE_SITE"] = str(site)
    import torch
    sys.path.insert(0, ".")
    from diffusion_model_nemo_amd import _native as N
    from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    u = Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
             compute_dtype="bf16").to(dev).eval()
    gd = GaussianDiffusion(timesteps=40, schedule_name="cosine")
    gd.p_sample_loop(u, (64, 3, 64, 64), device=dev, seed=3, keep="last", return_device_tensor=True)
    h = u.native_handle()
    N.check(N.lib().dmn_unet_probe_dominant(h, 2))
    gd.p_sample_loop(u, (64, 3, 64, 64), device=dev, seed=3, keep="last", return_device_tensor=True)
    torch.cuda.synchronize()
    n = 16384
    buf = np.zeros((n, 8), np.int64)
    N.check(N.lib().dmn_test_probe_read(h, 0, buf.ctypes.data_as(C.c_void_p), n))
    N.check(N.lib().dmn_unet_probe_dominant(h, 0))
    d = buf[buf[:, 0] != 0]
    if len(d) == 0:
        print(f"site {site}: no stamps (not a convolution launch with stamps, or beyond the last)")
        return
    ph = np.stack([d[:, 1] - d[:, 7], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 4] - d[:, 3]], 1)
    med = np.median(ph, 0)
    life = (d[:, 5] - d[:, 0]) / 100.0
    span = (d[:, 5].max() - d[:, 0].min()) / 100.0
    print(f"site {site:2d}: wgs={len(d):5d} span={span:6.1f}us life med={np.median(life):5.1f}us  cycles: setup={med[0]:7.0f} "
          f"stage={med[1]:7.0f} mfma={med[2]:7.0f} epilogue={med[3]:7.0f}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--one":
        one(int(sys.argv[2]))
    else:
        for s in sys.argv[1:]:
            subprocess.run([sys.executable, __file__, "--one", s], check=False)
