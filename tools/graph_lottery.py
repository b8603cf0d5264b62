"""Experiment (GPU): does the device time of a replayed step depend on the graph INSTANCE? One handle, one image buffer:
(a) the same captured graphs replayed in several loops, (b) the graphs captured anew before every loop (a second image
buffer alternates with the first, which changes the key the graphs are cached under).
python tools/graph_lottery.py [steps]"""
import sys

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402
from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
u = Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
         compute_dtype="bf16").to(dev).eval()
gd = GaussianDiffusion(timesteps=steps, schedule_name="cosine")
xa = torch.empty((64, 3, 64, 64), device=dev)
xb = torch.empty_like(xa)
h = u.native_handle()


def loop(x):
    gd._run_chain(u, x, 0, steps, True, "last", 3, 0)
    torch.cuda.synchronize()
    return N.loop_last_ms(h) / steps * 1e3


loop(xa)
for trial in range(5):
    same = [loop(xa) for _ in range(4)]           # first of these captures (the key changed), the rest replay that instance
    print(f"LOTTERY instance {trial}: {' '.join(f'{m:.1f}' for m in same)} us per step", flush=True)
    loop(xb)                                      # another key: the next loop on xa captures again
u.check_handoffs()
