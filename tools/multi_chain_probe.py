"""Experiment (GPU): N independent single-chain handles sampling side by side on N streams (one host thread each), to
price a third launch chain before building it. usage: python tools/multi_chain_probe.py <batches e.g. 32,32 or 22,21,21> [steps]"""
import os
import sys
import threading
import time

os.environ["DMN_SAMPLE_CHAINS"] = "1"
import torch  # noqa: E402

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402
from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet  # noqa: E402

batches = [int(b) for b in sys.argv[1].split(",")]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
units = []
for b in batches:
    u = Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
             compute_dtype="bf16").to(dev).eval()
    units.append((u, GaussianDiffusion(timesteps=steps, schedule_name="cosine"), torch.cuda.Stream(dev), b))


def run(u, gd, st, b):
    with torch.cuda.stream(st):
        gd.p_sample_loop(u, (b, 3, 64, 64), device=dev, seed=3, keep="last", return_device_tensor=True, use_graph=False)


for x in units:   # warm-up one at a time (workspaces, handle state)
    run(*x)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=x) for x in units]
    for h in th:
        h.start()
    for h in th:
        h.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"batches {batches} mask {os.environ.get('DMN_UNET_PATHS', '0')}: rep {rep}: {dt / steps * 1e6:.1f} us per step "
          f"({sum(batches) / dt * steps / 1000:.1f} img/s at 1000 steps)", flush=True)
