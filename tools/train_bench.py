"""Times the DDPM training step (q_sample + U-Net forward + L_simple + backward + clip + AdamW) of the benchmark
configuration 5: 128x128 images, 32 per GPU, shipped network widths. `python tools/train_bench.py [--batch 32]
[--size 128] [--steps 20] [--warmup 5]`; under torch.distributed.run every rank takes `--batch` images and the flat
gradient buffer is all-reduced over RCCL. Prints one JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"], help="matrix-core operand type")
    ap.add_argument("--activations", default=None, choices=["bf16", "fp32"], help="storage of saved activations")
    ap.add_argument("--extra-streams", type=int, default=0, help="streams created and used before the trainer's own (queue-sharing study)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
    from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet
    from diffusion_model_nemo_amd.training import UnetTrainer, cosine_annealing_lr
    extra = [torch.cuda.Stream() for _ in range(a.extra_streams)]
    for st in extra:
        with torch.cuda.stream(st):
            torch.zeros(16, device="cuda").add_(1)
    torch.cuda.synchronize()
    torch.manual_seed(1234)
    u = Unet(input_dim=a.size, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False,
             dropout=0.0, compute_dtype=a.dtype).cuda()
    tr = UnetTrainer(u, "l2", activations=a.activations)
    gd = GaussianDiffusion(timesteps=1000, schedule_name="cosine")
    sa = gd.sqrt_alphas_cumprod.cuda()
    sb = gd.sqrt_one_minus_alphas_cumprod.cuda()
    gen = torch.Generator(device="cuda").manual_seed(local)
    x0 = torch.rand(a.batch, 3, a.size, a.size, device="cuda", generator=gen) * 2 - 1

    def step(i):
        t = torch.randint(0, 1000, (a.batch,), device="cuda", generator=gen)
        eps = torch.randn(a.batch, 3, a.size, a.size, device="cuda", generator=gen)
        loss = tr.loss_and_grads(x0, t, eps, sa, sb)
        scale = tr.all_reduce_grads()
        tr.optimizer_step(lr=cosine_annealing_lr(i, 10000, 1e-3, 1e-4), betas=(0.9, 0.98), weight_decay=1e-3,
                          max_grad_norm=1.0, grad_scale=scale)
        return loss

    for i in range(a.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = (time.perf_counter() - t0) / a.steps
    flop_img = {32: 0.6659e9, 64: 2.6634e9, 128: 10.677e9}.get(a.size, 10.677e9 * (a.size / 128) ** 2) * 3
    if int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({"metric": "images/sec, DDPM training step", "value": a.batch * world / dt, "unit": "images/sec",
                          "n_gpus": world, "ms_per_step": dt * 1e3, "batch_per_gpu": a.batch, "size": a.size,
                          "tflops_per_gpu": a.batch * flop_img / dt / 1e12, "dtype": {"fp32": "f32", "bf16": "bf16 operands, f32 accumulate, %s activations" % tr.activations}[a.dtype], "loss": float(loss)}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
