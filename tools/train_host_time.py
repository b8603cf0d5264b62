"""Host time spent enqueueing one training step (no synchronisation inside the timed region) next to the device time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet
from diffusion_model_nemo_amd.training import UnetTrainer
u = Unet(input_dim=128, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False, dropout=0.0).cuda()
tr = UnetTrainer(u, "l2")
gd = GaussianDiffusion(timesteps=1000, schedule_name="cosine")
sa, sb = gd.sqrt_alphas_cumprod.cuda(), gd.sqrt_one_minus_alphas_cumprod.cuda()
x0 = torch.rand(32, 3, 128, 128, device="cuda") * 2 - 1
t = torch.randint(0, 1000, (32,), device="cuda")
eps = torch.randn_like(x0)
for _ in range(3):
    tr.loss_and_grads(x0, t, eps, sa, sb)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(10):
    a = time.perf_counter()
    tr.loss_and_grads(x0, t, eps, sa, sb)
    host.append(time.perf_counter() - a)
torch.cuda.synchronize()
total = (time.perf_counter() - t0) / 10
print("host enqueue ms/step", [round(h * 1e3, 2) for h in host], "wall ms/step", round(total * 1e3, 2))
