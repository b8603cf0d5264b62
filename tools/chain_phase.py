"""Tuning aid (GPU): how the two launch chains of a sampling loop run against each other. The opening kernel of every
network evaluation stamps the device clock (dmn_test_step_trace); per loop this prints the device time per step, each
chain's mean step duration, and the phase of chain 1 inside chain 0's step (start of chain 1's step i minus start of
chain 0's step i, as a fraction of a step) at several points of the loop.
python tools/chain_phase.py [steps] [loops]"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402
from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
loops = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
u = Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
         compute_dtype="bf16").to(dev).eval()
gd = GaussianDiffusion(timesteps=steps, schedule_name="cosine")
x = torch.empty((64, 3, 64, 64), device=dev)
h = u.native_handle()
gd._run_chain(u, x, 0, steps, True, "last", 3, 0)
torch.cuda.synchronize()
for loop in range(loops):
    N.check(N.lib().dmn_test_step_trace(h, 0, None))
    gd._run_chain(u, x, 0, steps, True, "last", 3, 0)
    torch.cuda.synchronize()
    us = N.loop_last_ms(h) / steps * 1e3
    ts = []
    for c in range(2):
        buf = np.zeros(8192, np.int64)
        N.check(N.lib().dmn_test_step_trace(h, c, buf.ctypes.data_as(C.c_void_p)))
        t = np.sort(buf[buf != 0]) / 100.0   # microseconds
        ts.append(t)
    n = min(len(ts[0]), len(ts[1]))
    if n < steps // 2:
        print(f"PHASE loop {loop}: {us:.1f} us per step, stamps {len(ts[0])} / {len(ts[1])}")
        continue
    d0, d1 = np.diff(ts[0][:n]), np.diff(ts[1][:n])
    lag = ts[1][:n] - ts[0][:n]
    step = d0.mean()
    marks = [0, n // 8, n // 4, n // 2, 3 * n // 4, n - 1]
    print(f"PHASE loop {loop}: {us:.1f} us per step; chain step {d0.mean():.1f} / {d1.mean():.1f} us (std {d0.std():.1f} / {d1.std():.1f}); "
          f"lag of chain 1 in steps at {marks}: " + " ".join(f"{lag[m] / step:+.2f}" for m in marks), flush=True)
u.check_handoffs()
