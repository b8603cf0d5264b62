#!/usr/bin/env python3
"""Benchmark of the north-star workload: images/sec of 1000-step DDPM sampling with the 64x64x3 U-Net
(dim=32, dim_mults=[1,2,4,8], 8.08 M parameters, random weights, cosine schedule), bf16 internal dtype,
batch 64 per GPU.

    python bench.py --gpus N --steps K --warmup W

One "step" is one complete 1000-step sampling chain of one 64-image batch per GPU (x_T drawn on the device,
1000 x {time MLP, U-Net, posterior update} replayed as a hipGraph, final images gathered over RCCL when N > 1
and copied to the host). Prints ONE JSON line on rank 0.

`roofline` describes the kernel that takes the largest share of a denoising step (the level-0 3x3 convolution
with fused GroupNorm + SiLU prologue that also closes its ResnetBlock): its duration is measured here by the
launch's own workgroups (device clock stamps inside the graph-replayed steps), its HBM traffic comes from the
rocprofv3 counter passes summarised in profiles/ (null when that file is absent).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Process-wide setting of the HIP runtime, read when it initialises (so before torch is imported): graph launches
# enqueue their kernel nodes through the ordinary dispatch path instead of replaying packets recorded at
# instantiation. Measured on MI355X with this loop: 708.0-708.5 against 711.6-711.9 us per denoising step (-0.5 %);
# results are unaffected. A caller's own setting wins; the line printed below reports what was in force.
RUNTIME_ENV = {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}
for _k, _v in RUNTIME_ENV.items():
    os.environ.setdefault(_k, _v)

import torch  # noqa: E402

# work per image per network evaluation at 64x64 (measured with hooks on the reference network; MAC = 2 FLOP,
# bf16 activations, each conv / linear / attention-matmul input read once and output written once)
GFLOP_PER_IMG_EVAL = 2.6634
ALG_BYTES_PER_IMG_EVAL = 17.3e6
WEIGHT_BYTES = 16.2e6
HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp32": 157.3}   # dense matrix peaks of the two compute types
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "r03_dominant_kernel_traffic.json")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--image-size", type=int, default=64)
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the extra training-step measurement")
    ap.add_argument("--learned-variance", action="store_true",
                    help="improved DDPM: 6-channel network + LearnedGaussianDiffusion (benchmark configuration 3 per GPU)")
    ap.add_argument("--probe-steps", type=int, default=100, help="eager steps for the in-loop timing of the dominant kernel")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for --gpus > 1 (nccl is RCCL on ROCm; gloo only together with --dry-run)")
    ap.add_argument("--dry-run", action="store_true",
                    help="rehearsal of the multi-rank control flow on CPU tensors: the sampler is replaced by one that fills every "
                         "image with its global index and the training step by a fixed gradient, so process-group set-up, image "
                         "offsets, the gather, the agreement reduce of the training extra, the timing reduce and the JSON line "
                         "can be exercised without a GPU (the number it prints measures nothing)")
    a = ap.parse_args(argv)
    if a.backend == "gloo" and not a.dry_run:
        ap.error("--backend gloo moves no device tensors here: use it with --dry-run")
    return a


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_chain(B, image_size, timesteps, budget_s, learned_variance=False, min_steps=3):
    """p_sample steps of the CPU oracle (a pinned port of the reference path) on this host until `timesteps` steps or
    `budget_s` seconds have run; returns (steps done, seconds)."""
    from oracle import philox, ref_cpu as R
    from oracle.weights import make_state_dict, unet_param_shapes
    p = R.to_torch(make_state_dict(unet_param_shapes(learned_variance=learned_variance)))
    c = R.ddpm_constants(timesteps, "cosine")
    shape = (B, 3, image_size, image_size)
    x = torch.from_numpy(philox.normal_like(shape, 0, timesteps, 0))
    n, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while True:
            i = timesteps - 1 - n
            t = torch.full((B,), i, dtype=torch.long)
            eps = R.unet_forward(p, x, t)
            x = R.p_sample_update(c, x, t, eps, torch.from_numpy(philox.normal_like(shape, 0, i, 1)),
                                  learned_variance=learned_variance)
            n += 1
            el = time.perf_counter() - t0
            if (el >= budget_s and n >= min_steps) or n >= timesteps:
                break
    return n, el


def cpu_baseline(batch, image_size, timesteps, budget_s, learned_variance=False):
    """The CPU oracle timed on this host's cores: (1) the timed workload itself (same batch, image size and step count)
    for a bounded number of steps, extrapolated to the whole chain; (2) benchmark configuration 1 (32x32, batch 4, the
    reference's own CPU-runnable case) for up to 1000 steps."""
    cores = min(16, os.cpu_count() or 1)   # the box's CPU share for one GPU is 16 threads, whatever the machine has
    torch.set_num_threads(cores)
    n, el = cpu_chain(batch, image_size, timesteps, budget_s, learned_variance)
    sps = el / n
    n1, el1 = cpu_chain(4, 32, 1000, 2.5 * budget_s)
    c1 = {"value": 4 / (el1 / n1 * 1000), "unit": "images/sec",
          "sample": f"32x32, batch 4, {n1} of 1000 p_sample steps in {el1:.1f}s"
                    + ("" if n1 == 1000 else f" ({el1 / n1 * 1e3:.1f} ms/step), extrapolated x1000")}
    return {"value": batch / (sps * timesteps), "unit": "images/sec", "cores": cores, "kind": "port",
            "cpu_model": cpu_model_name(), "host_logical_cpus": os.cpu_count(),
            "sample": f"CPU oracle, batch {batch} {image_size}x{image_size}, {n} p_sample steps in {el:.1f}s "
                      f"({sps * 1e3:.1f} ms/step), extrapolated x{timesteps}",
            "config1_32x32_batch4": c1}


def _dry_train_step(dev, rank, world):
    """Stand-in for the training step in --dry-run: a fixed per-rank gradient through the same all-reduce helper."""
    from diffusion_model_nemo_amd.parallel import all_reduce_flat_grads

    def step(i, reduce):
        grads = torch.full((1024,), float(rank + 1), device=dev)
        scale = all_reduce_flat_grads(grads) if reduce else 1.0
        if reduce:   # the mean over the ranks of (rank + 1)
            assert abs(float(grads[0]) * scale - (world + 1) / 2.0) < 1e-6, (float(grads[0]), scale)
        return torch.tensor(1.0 / (i + 1), device=dev)

    return step


def _real_train_step(dev, rank, activations=None):
    from diffusion_model_nemo_amd.modules import GaussianDiffusion, Unet
    from diffusion_model_nemo_amd.training import UnetTrainer, cosine_annealing_lr
    Bt, St = 32, 128
    net = Unet(input_dim=St, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8, use_convnext=False,
               dropout=0.0, compute_dtype="bf16").to(dev)
    tr = UnetTrainer(net, "l2", activations=activations)
    gd = GaussianDiffusion(timesteps=1000, schedule_name="cosine")
    sa, sb = gd.sqrt_alphas_cumprod.to(dev), gd.sqrt_one_minus_alphas_cumprod.to(dev)
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    x0 = torch.rand(Bt, 3, St, St, device=dev, generator=gen) * 2 - 1

    def step(i, reduce):
        t = torch.randint(0, 1000, (Bt,), device=dev, generator=gen)
        eps = torch.randn(Bt, 3, St, St, device=dev, generator=gen)
        loss = tr.loss_and_grads(x0, t, eps, sa, sb)
        scale = tr.all_reduce_grads() if reduce else 1.0
        tr.optimizer_step(lr=cosine_annealing_lr(i, 10000, 1e-3, 1e-4), betas=(0.9, 0.98), weight_decay=1e-3,
                          max_grad_norm=1.0, grad_scale=scale)
        return loss

    return step


def train_step_extra(dev, rank, world, dist, dry=False):
    """Extra, reported beside the headline: the DDPM training step of benchmark configuration 5 (128x128, 32 images per
    GPU: q_sample + forward + L_simple + backward, ONE all-reduce of the flat gradient buffer over RCCL when N > 1,
    global-norm clip + AdamW), bf16 matrix-core operands, fp32 activations. Every rank first runs one step without any
    collective and the ranks agree that all succeeded before the timed loop with the all-reduce starts, so a failure
    here cannot hang the job; it only drops this entry."""
    ok, step = 1, None

    def sync():
        if not dry:
            torch.cuda.synchronize(dev)

    try:
        if dry:
            step = _dry_train_step(dev, rank, world)
            if os.environ.get("DMN_BENCH_DRY_FAIL_RANK") == str(rank):   # rehearsal of a rank that cannot run the extra
                raise RuntimeError("injected failure")
        else:
            step = _real_train_step(dev, rank)
        first = float(step(0, False))
        ok = 1 if first == first else 0   # NaN check
    except Exception as e:   # noqa: BLE001 - the headline number must survive a failure of the extra
        print(f"[bench] training-step extra skipped on rank {rank}: {e}", file=sys.stderr)
        ok = 0
    flag = torch.tensor([ok], device=dev, dtype=torch.int32)
    if dist is not None:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        return None
    warm, iters = (1, 2) if dry else (4, 20)
    for i in range(warm):
        step(1 + i, True)
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for i in range(iters):
        loss = step(1 + warm + i, True)
    if dist is not None:
        dist.barrier()
    sync()
    tt = torch.tensor([(time.perf_counter() - t0) / iters], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    other = None
    if dist is None and not dry:
        # the same step with the saved activations STORED in bf16 (same gradient band as fp32 storage in the training tests): reported
        # beside the default mode, single GPU only
        try:
            del step
            step2 = _real_train_step(dev, rank, activations="bf16")
            for i in range(1 + warm):
                loss2 = step2(i, True)
            sync()
            t0 = time.perf_counter()
            for i in range(iters):
                loss2 = step2(1 + warm + i, True)
            sync()
            dt2 = (time.perf_counter() - t0) / iters
            other = {"value": 32 / dt2, "unit": "images/sec", "ms_per_step": dt2 * 1e3, "loss": float(loss2),
                     "dtype": "bf16 matrix-core operands and activation storage, fp32 accumulate / optimiser"}
        except Exception as e:   # noqa: BLE001
            print(f"[bench] bf16-activation training step skipped: {e}", file=sys.stderr)
    return {"bf16_activation_storage": other,
            "metric": "images/sec, DDPM training step (q_sample + fwd + L_simple + bwd + grad all-reduce + clip + AdamW)",
            "value": 32 * world / dt, "unit": "images/sec", "ms_per_step": dt * 1e3, "n_gpus": world,
            "config": {"workload": "128x128x3 DDPM U-Net (dim 32, mults 1-2-4-8), 32 images per GPU", "global_batch": 32 * world,
                       "parallelism": f"dp{world}, one all-reduce of 32.3 MB fp32 gradients per step"},
            "dtype": "bf16 matrix-core operands, fp32 accumulate / activations / optimiser", "scaling": "weak",
            "tflops_per_gpu": 32 * 3 * 10.677e9 / dt / 1e12, "loss": float(loss)}


def main(argv=None):
    a = parse(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dry = a.dry_run
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dev = torch.device("cpu") if dry else torch.device("cuda", local)
    if not dry:
        torch.cuda.set_device(dev)

    from diffusion_model_nemo_amd.parallel import gather_images
    B, S = a.batch, a.image_size
    shape = (B, 3, S, S)
    unet = gd = N = None
    if not dry:
        from diffusion_model_nemo_amd import _native as N
        from diffusion_model_nemo_amd.modules import GaussianDiffusion, LearnedGaussianDiffusion, Unet
        torch.manual_seed(1234)
        unet = Unet(input_dim=a.image_size, dim=32, dim_mults=[1, 2, 4, 8], channels=3, resnet_block_groups=8,
                    use_convnext=False, dropout=0.0, learned_variance=a.learned_variance, compute_dtype=a.dtype).to(dev).eval()
        gd = (LearnedGaussianDiffusion if a.learned_variance else GaussianDiffusion)(timesteps=a.timesteps, schedule_name="cosine")

    def local_images(k, image_offset):
        if dry:   # every image carries its global index: the gather below can be checked exactly
            return (torch.arange(B, dtype=torch.float32) + image_offset).view(B, 1, 1, 1).expand(*shape).contiguous()
        # the call waits for the chain and raises if an in-launch exchange of GroupNorm sums timed out
        return gd.p_sample_loop(unet, shape, device=dev, seed=1000 + k, image_offset=image_offset, return_device_tensor=True,
                                use_graph=not a.no_graph)

    def one_sample(k):
        x0 = local_images(k, rank * B)
        x0 = gather_images(x0) if world > 1 else x0
        return x0.cpu() if rank == 0 else x0   # final images on the host, as the reference returns them

    def fence():
        if dist is not None:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize(dev)

    for k in range(a.warmup):
        one_sample(-1 - k)
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        out = one_sample(k)
    fence()
    el = time.perf_counter() - t0
    assert torch.isfinite(out).all()
    assert out.shape[0] == world * B, (tuple(out.shape), world, B)
    if dry:
        assert torch.equal(out[:, 0, 0, 0], torch.arange(world * B, dtype=torch.float32)), "gather order / image offsets"
    tmax = torch.tensor([el], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    el = float(tmax.item())
    # in-launch exchanges that ran into their time limit, over all ranks (p_sample_loop has already raised on the rank
    # concerned; the count is reported so that a reader of the line sees it was looked at)
    gv = torch.tensor([0 if dry else unet.handoff_giveups()], device=dev, dtype=torch.int64)
    if dist is not None:
        dist.all_reduce(gv, op=dist.ReduceOp.SUM)
    giveups = int(gv.item())

    if dry:
        train = None if a.no_train else train_step_extra(dev, rank, world, dist, dry=True)
        if rank == 0:
            res = {"metric": "images/sec, 1000-step DDPM sample, 64x64 U-Net", "value": world * B * a.steps / el, "unit": "images/sec",
                   "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": el / a.steps * 1e3,
                   "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
                   "data": "none: dry run of the control flow (index-valued images, fixed gradients); the value measures nothing",
                   "dry_run": True, "backend": a.backend if world > 1 else None, "handoff_giveups": giveups,
                   "config": {"workload": f"dry run, batch {B} per rank", "global_batch": world * B, "parallelism": f"dp{world}"}}
            if train is not None:
                res["train_step"] = train
            print(json.dumps(res), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # device time of the last chain, from HIP events recorded on the loop's own stream
    loop_ms = N.loop_last_ms(unet.native_handle())

    # The dominant kernel where it runs: HIP event pairs around its launch inside an eager (launch by launch) replay of
    # the same step sequence on the stream the loop uses, so it follows and precedes the same kernels as in the captured
    # step. Beside it, the same kernel launched back to back on one tensor (warm caches), HIP events on its stream.
    import ctypes as C
    h = unet.native_handle()
    probe_steps = max(10, min(a.probe_steps, 250))
    gdp = type(gd)(timesteps=probe_steps, schedule_name="cosine")
    pair_ms, mark_ms, p_n, k_bytes = C.c_float(), C.c_float(), C.c_int(), C.c_double()
    # (a) device clock: the launch's workgroups stamp the 100 MHz clock at start and end; works inside the captured steps,
    #     so the launch is timed where the timed chains run it (graph replay, both launch chains in flight)
    N.check(N.lib().dmn_unet_probe_dominant(h, 2))
    gdp.p_sample_loop(unet, shape, device=dev, seed=7, keep="last", return_device_tensor=True, use_graph=not a.no_graph)
    clk_ms, clk_n = C.c_float(), C.c_int()
    N.check(N.lib().dmn_unet_probe_result(h, C.byref(clk_ms), C.byref(mark_ms), C.byref(clk_n), C.byref(k_bytes)))
    # (a') the same with ONE launch chain (64 images per launch, nothing else in flight): the kernel by itself in its
    #      place in the step, comparable with earlier rounds and with a profiler's per-kernel average
    one_ms, one_n, one_bytes = C.c_float(), C.c_int(), C.c_double()
    chains_default = N.lib().dmn_ddpm_loop_chains(h, B)
    if chains_default > 1:
        unet.set_sample_chains(1)
        N.check(N.lib().dmn_unet_probe_dominant(unet.native_handle(), 2))
        gdp.p_sample_loop(unet, shape, device=dev, seed=7, keep="last", return_device_tensor=True, use_graph=not a.no_graph)
        N.check(N.lib().dmn_unet_probe_result(h, C.byref(one_ms), C.byref(mark_ms), C.byref(one_n), C.byref(one_bytes)))
        unet.set_sample_chains(2)
    # (a'') the matrix-bound kernel: the 256 -> 256 channel 3x3 convolution at 8x8 (the network runs several per
    #       evaluation), clocked the same way where it runs; priced against the dense bf16 / f32 MFMA peak
    mfma_sites = []
    want_flops = 2.0 * (B // max(chains_default, 1)) * 8 * 8 * 256 * 256 * 9
    if S == 64:
        for site in range(10, 30):
            N.check(N.lib().dmn_test_probe_site(h, site, None))
            N.check(N.lib().dmn_unet_probe_dominant(h, 2))
            gdp.p_sample_loop(unet, shape, device=dev, seed=7, keep="last", return_device_tensor=True, use_graph=not a.no_graph)
            s_ms, s_n, s_fl = C.c_float(), C.c_int(), (C.c_double * 5)()
            if N.lib().dmn_unet_probe_result(h, C.byref(s_ms), C.byref(mark_ms), C.byref(s_n), None) != 0:
                continue   # no stamping launch at this number
            N.check(N.lib().dmn_test_probe_site(h, site, s_fl))
            if list(s_fl) == [want_flops, 256.0, 256.0, 8.0, 9.0]:
                mfma_sites.append((site, s_ms.value * 1e3, s_n.value))
        N.check(N.lib().dmn_test_probe_site(h, -1, None))
    # (b) HIP events around the launch inside an eager replay of the same steps (markers cannot be timed inside a graph)
    N.check(N.lib().dmn_unet_probe_dominant(h, 1))
    gdp.p_sample_loop(unet, shape, device=dev, seed=7, keep="last", return_device_tensor=True, use_graph=False)
    N.check(N.lib().dmn_unet_probe_dominant(h, 0))
    N.check(N.lib().dmn_unet_probe_result(h, C.byref(pair_ms), C.byref(mark_ms), C.byref(p_n), C.byref(k_bytes)))
    p_ms = C.c_float(pair_ms.value - mark_ms.value)   # the kernel without the marker packet that closes the pair
    k_ms, kb2 = C.c_float(), C.c_double()
    ws = unet.workspace(B, S, S, dev, sampling=True)
    stream = torch.cuda.current_stream(dev).cuda_stream
    chains = N.lib().dmn_ddpm_loop_chains(h, B)   # the loop runs the batch as this many launch chains
    Bl = B // chains                              # images per launch
    N.check(N.lib().dmn_bench_dominant_kernel(h, Bl, S, S, 200, ws.data_ptr(), ws.numel(), stream,
                                              C.byref(k_ms), C.byref(kb2)))
    traffic = None
    if os.path.exists(TRAFFIC_FILE):
        with open(TRAFFIC_FILE) as fh:
            tj = json.load(fh)
        if tj.get("batch") == Bl and tj.get("image_size") == S and tj.get("dtype") == a.dtype:
            traffic = tj["hbm_bytes_per_launch"]

    train = None if a.no_train else train_step_extra(dev, rank, world, dist)

    # what this device sustains, beside the nominal peaks the roofline is priced against (rank 0, single GPU only)
    peaks = None
    if rank == 0 and world == 1:
        from diffusion_model_nemo_amd.peaks import measure as measure_peaks
        pk = measure_peaks(dev)
        big = max(e["bytes"] for e in pk["stream"])
        small = min(e["bytes"] for e in pk["stream"])
        rate = {(e["kind"], e["bytes"]): e["GBs"] for e in pk["stream"]}
        peaks = {"hbm_copy_GBs": rate[("copy", big)], "hbm_read_GBs": rate[("read", big)],
                 "copy_of_one_level0_activation_GBs": rate[("copy", small)],
                 "mfma_bf16_TFLOPs": max(m["TFLOPs"] for m in pk["mfma"] if "bf16" in m["instruction"]),
                 "mfma_f32_TFLOPs": max(m["TFLOPs"] for m in pk["mfma"] if "bf16" not in m["instruction"])}
        del pk
        torch.cuda.empty_cache()

    if rank == 0:
        imgs = world * B * a.steps
        sec_per_eval = (loop_ms / 1e3) / a.timesteps
        alg_bytes = B * ALG_BYTES_PER_IMG_EVAL + WEIGHT_BYTES
        res = {
            "metric": "images/sec, 1000-step DDPM sample, 64x64 U-Net" + (" (improved DDPM, learned variance)" if a.learned_variance else ""),
            "value": imgs / el, "unit": "images/sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic (Philox x_T and noise, random-init weights)", "handoff_giveups": giveups,
            "config": {"workload": f"{S}x{S}x3 {'improved-DDPM (6-channel head)' if a.learned_variance else 'DDPM'} U-Net (dim 32, mults 1-2-4-8), {a.timesteps}-step p_sample_loop, "
                                   f"batch {B} per GPU", "global_batch": world * B, "parallelism": f"dp{world}",
                       "graph": not a.no_graph, "launch_chains": chains,
                       "runtime_env": {k: os.environ.get(k) for k in RUNTIME_ENV}},
            "roofline": {"bound": "hbm", "achieved": k_bytes.value / (clk_ms.value / 1e3) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": k_bytes.value / (clk_ms.value / 1e3) / 1e9 / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "kernel": "conv_igemm_kernel<bf16, 3x3, 32x16 px, 32 cout, GN+SiLU prologue, block tail inside the launch> "
                                   f"[{Bl},{S},{S},32] (+ residual [{Bl},{S},{S},32]) -> [{Bl},{S},{S},32] per launch "
                                   f"({chains} chain{'s' if chains > 1 else ''} of {Bl} images): block2.proj of downs.0.0, which also "
                                   "closes its ResnetBlock (GroupNorm partial sums exchanged between the image's workgroups)",
                         "algorithmic_bytes": k_bytes.value, "launch_us": clk_ms.value * 1e3,
                         "launch_us_measured": f"device clock (100 MHz) stamps of the launch's workgroups, earliest start to latest "
                                               f"end, {clk_n.value} launches inside graph-replayed denoising steps with "
                                               f"{chains} launch chain{'s' if chains > 1 else ''} in flight; the end stamp waits "
                                               "for the workgroup's stores. A profiler's dispatch duration of the same launch is "
                                               "2-3 us longer: dispatch and completion (1.6 us for any kernel here) and the "
                                               "write-back of the 8.4 MB the launch leaves in L2 when it ends",
                         "note": "with two launch chains in flight a launch shares the machine with the other chain's kernels: its own "
                                 "bytes over its own duration understate what the machine moves meanwhile; single_chain is the same "
                                 "kernel timed the same way with one chain (nothing else in flight; a launch whose workgroups "
                                 "exchange takes at most 256 of them, i.e. 32 images: one chain issues two such launches, the "
                                 "figures are those of one)",
                         "single_chain": None if chains_default == 1 else {
                             "launch_us": one_ms.value * 1e3, "algorithmic_bytes": one_bytes.value, "launches": one_n.value,
                             "achieved": one_bytes.value / (one_ms.value / 1e3) / 1e9,
                             "frac": one_bytes.value / (one_ms.value / 1e3) / 1e9 / HBM_PEAK_GBS},
                         "launch_us_hip_events_eager": p_ms.value * 1e3,
                         "launch_us_hip_events_eager_measured": f"HIP events around {p_n.value} launches inside eager denoising steps: marker-to-"
                                               f"marker {pair_ms.value * 1e3:.2f} us minus {mark_ms.value * 1e3:.2f} us for a marker by itself",
                         "back_to_back": {"what": "the same convolution WITHOUT the block tail (16x16-pixel tiles, input + output + "
                                                  "weights), launched back to back on one tensor; HIP events on its stream",
                                          "launch_us": k_ms.value * 1e3, "algorithmic_bytes": kb2.value,
                                          "frac": kb2.value / (k_ms.value / 1e3) / 1e9 / HBM_PEAK_GBS}},
            # the whole captured step (time MLP + U-Net + posterior update) against the same roof
            "step": {"device_ms": sec_per_eval * 1e3, "algorithmic_bytes": alg_bytes,
                     "achieved_GBs": alg_bytes / sec_per_eval / 1e9,
                     "frac_of_hbm_peak": alg_bytes / sec_per_eval / 1e9 / HBM_PEAK_GBS,
                     "tflops": B * GFLOP_PER_IMG_EVAL / sec_per_eval / 1e3},
        }
        if mfma_sites:
            peak_tf = MFMA_PEAK_TFLOPS[a.dtype]
            us = sum(m[1] for m in mfma_sites) / len(mfma_sites)
            res["roofline"]["mfma"] = {
                "bound": "mfma", "kernel": f"conv_igemm_kernel<{a.dtype}, 3x3, 8x8 px, 32 cout> [{Bl},8,8,256] -> [{Bl},8,8,256] per launch",
                "flops_per_launch": want_flops, "launch_us": us, "achieved": want_flops / (us * 1e-6) / 1e12, "peak": peak_tf,
                "unit": "TFLOP/s", "frac": want_flops / (us * 1e-6) / 1e12 / peak_tf,
                "launches_per_evaluation": len(mfma_sites),
                "launch_us_by_site": {str(m[0]): round(m[1], 2) for m in mfma_sites},
                "launch_us_measured": f"device clock stamps as above, {mfma_sites[0][2]} launches per site inside graph-replayed "
                                      f"denoising steps with {chains} launch chain{'s' if chains > 1 else ''} in flight"}
        if peaks is not None:
            res["measured_peaks"] = peaks
            res["roofline"]["frac_of_measured_hbm_copy"] = res["roofline"]["achieved"] / peaks["hbm_copy_GBs"]
        if train is not None:
            res["train_step"] = train
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(B, S, a.timesteps, a.cpu_seconds, a.learned_variance)
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
