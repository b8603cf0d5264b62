"""Experiment (GPU): one 64-image chain vs. two (four) concurrent 32- (16-) image chains on separate streams."""
import sys
import time

import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet  # noqa: E402

dev = torch.device("cuda", 0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def build():
    torch.manual_seed(1234)
    return Unet(input_dim=64, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False,
                dropout=0.0, compute_dtype="bf16").to(dev).eval()


def run(nsplit):
    B = 64 // nsplit
    nets = [build() for _ in range(nsplit)]
    gds = [GaussianDiffusion(timesteps=T, schedule_name="cosine") for _ in range(nsplit)]
    streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
    outs = [None] * nsplit

    def once():
        for i in range(nsplit):
            with torch.cuda.stream(streams[i]):
                outs[i] = gds[i].p_sample_loop(nets[i], (B, 3, 64, 64), device=dev, seed=7, image_offset=i * B,
                                               return_device_tensor=True)
        torch.cuda.synchronize()

    once()
    t0 = time.perf_counter()
    once()
    el = time.perf_counter() - t0
    print(f"{nsplit} x B={B}: {el * 1e3 / T:.3f} ms per step for 64 images ({64 / (el * 1000 / T):.1f} img/s at 1000 steps)", flush=True)
    return torch.cat(outs)


a = run(1)
b = run(2)
c = run(4)
print("max |1x - 2x| =", float((a - b).abs().max()), " max |1x - 4x| =", float((a - c).abs().max()))
