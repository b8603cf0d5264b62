"""Experiment (GPU): the training step's forward + backward (about 1900 launches on four streams) captured once into a
graph and replayed, against the eager step: does the device run faster when the host is not enqueueing beside it?
python tools/train_graph_probe.py [--steps 20]"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet  # noqa: E402
from diffusion_model_nemo_amd.training import UnetTrainer, cosine_annealing_lr  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--size", type=int, default=128)
a = ap.parse_args()
torch.manual_seed(1234)
u = Unet(input_dim=a.size, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0,
         compute_dtype="bf16").cuda()
tr = UnetTrainer(u, "l2")
gd = GaussianDiffusion(timesteps=1000, schedule_name="cosine")
sa, sb = gd.sqrt_alphas_cumprod.cuda(), gd.sqrt_one_minus_alphas_cumprod.cuda()
gen = torch.Generator(device="cuda").manual_seed(0)
x0 = torch.rand(a.batch, 3, a.size, a.size, device="cuda", generator=gen) * 2 - 1
t = torch.randint(0, 1000, (a.batch,), device="cuda", generator=gen)
eps = torch.randn(a.batch, 3, a.size, a.size, device="cuda", generator=gen)
st = torch.cuda.Stream()


def opt(i):
    tr.optimizer_step(lr=cosine_annealing_lr(i, 10000, 1e-3, 1e-4), betas=(0.9, 0.98), weight_decay=1e-3, max_grad_norm=1.0,
                      grad_scale=tr.all_reduce_grads())


def timed(fn, tag):
    for i in range(5):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        fn(5 + i)
    torch.cuda.synchronize()
    print(f"TRAIN_GRAPH {tag}: {(time.perf_counter() - t0) / a.steps * 1e3:.3f} ms per step", flush=True)


with torch.cuda.stream(st):
    def eager(i):
        loss = tr.loss_and_grads(x0, t, eps, sa, sb)
        opt(i)
        return loss
    timed(eager, "eager")
    loss_eager = float(tr.loss_and_grads(x0, t, eps, sa, sb))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st, capture_error_mode="relaxed"):
        loss_g = tr.loss_and_grads(x0, t, eps, sa, sb)
    torch.cuda.synchronize()

    def replay(i):
        g.replay()
        opt(i)
    g.replay()
    torch.cuda.synchronize()
    print(f"TRAIN_GRAPH loss eager {loss_eager:.6f} graph {float(loss_g):.6f}", flush=True)
    timed(replay, "graph replay")
    timed(eager, "eager again")
