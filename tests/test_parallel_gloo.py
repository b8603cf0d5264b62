"""Batch-sharded sampling across processes (CPU, gloo, world size 2).

The GPU path shards a batch by image and gathers the finished images with one collective. Here the same helpers
(`diffusion_model_nemo_amd.parallel`) run with the CPU oracle as the per-rank sampler, for an even and for a ragged
split. The gather itself is checked bit for bit with an index-valued sampler; the oracle chain within 2e-5 abs,
because torch's CPU convolution picks batch-size dependent blockings (1-ulp differences between B=1, 2, 3, 4).
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STEPS = 4
SIDE = 16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_chain(batch, offset):
    """`batch` images starting at global index `offset`: a STEPS-step DDPM chain of the oracle U-Net."""
    from oracle import philox, ref_cpu as R
    from oracle.weights import make_state_dict, unet_param_shapes
    torch.set_num_threads(2)
    p = R.to_torch(make_state_dict(unet_param_shapes(dim=32, dim_mults=(1, 2))))
    c = R.ddpm_constants(STEPS, "linear")
    shape = (batch, 3, SIDE, SIDE)

    def noise_fn(step):   # stream 0: the x_T draw, stream 1: the per-step draws
        return torch.from_numpy(philox.normal_like(shape, 7, step, 0 if step == STEPS else 1, offset))

    def model_fn(x, t):
        return R.unet_forward(p, x, t, dim=32)

    with torch.no_grad():
        return R.ddpm_sample_chain(model_fn, c, shape, noise_fn)[0]


def _worker(rank, world, port, global_batch, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffusion_model_nemo_amd.parallel import shard_range, sharded_sample
        full = sharded_sample(_oracle_chain, global_batch)
        assert full.shape[0] == global_batch
        # exactness of the sharding + gather: every image carries its own global index
        idx = sharded_sample(lambda n, off: (torch.arange(n, dtype=torch.float32) + off).view(n, 1, 1, 1).expand(n, 3, 2, 2),
                             global_batch)
        assert torch.equal(idx[:, 0, 0, 0], torch.arange(global_batch, dtype=torch.float32))
        a, b = shard_range(global_batch, rank, world)
        assert 0 <= a < b <= global_batch
        if rank == 0:
            np.save(out_path, full.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("global_batch", [4, 3])
def test_two_rank_sharded_sampling_equals_single_process(tmp_path, global_batch):
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, _free_port(), global_batch, out), nprocs=2, join=True)
    got = np.load(out)
    want = _oracle_chain(global_batch, 0).numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-5


def test_shard_range_covers_batch_without_overlap():
    from diffusion_model_nemo_amd.parallel import shard_range
    for gb in (1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- data-parallel training: one all-reduce of the flat gradient buffer ------------------------------------------------
def _train_shapes():
    from oracle.weights import unet_param_shapes
    return unet_param_shapes(dim=32, dim_mults=(1, 2))


def _oracle_flat_grads(x0, t, noise, layout, flat_size):
    """Gradient of the mean L_simple loss over the given images, packed like the library's flat buffer."""
    from oracle import ref_cpu as R
    from oracle.weights import make_state_dict
    torch.set_num_threads(2)
    p = {k: torch.from_numpy(v.copy()).requires_grad_(True) for k, v in make_state_dict(_train_shapes()).items()}
    c = R.ddpm_constants(1000, "cosine")
    R.ddpm_training_loss(lambda x, tt: R.unet_forward(p, x, tt, dim=32), c, x0, t, noise).backward()
    flat = torch.zeros(flat_size)
    for name, (off, shape) in layout.items():
        flat[off:off + p[name].numel()] = p[name].grad.reshape(-1)
    return flat


def _train_inputs(global_batch):
    gen = torch.Generator().manual_seed(5)
    x0 = torch.tanh(torch.randn(global_batch, 3, SIDE, SIDE, generator=gen))
    noise = torch.randn(global_batch, 3, SIDE, SIDE, generator=gen)
    t = torch.randint(0, 1000, (global_batch,), generator=gen)
    return x0, t, noise


def _train_layout():
    from diffusion_model_nemo_amd import _native as N
    from diffusion_model_nemo_amd.training import flat_layout
    cfg = N.DmnUnetCfg()
    cfg.dim, cfg.n_mults, cfg.channels, cfg.out_dim, cfg.groups = 32, 2, 3, 3, 8
    cfg.dim_mults[0], cfg.dim_mults[1] = 1, 2
    cfg.heads, cfg.dim_head, cfg.tail_norm, cfg.with_time_emb = 4, 32, 1, 1
    return flat_layout(cfg)


def _train_worker(rank, world, port, global_batch, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffusion_model_nemo_amd.parallel import all_reduce_flat_grads, shard_range
        layout, n = _train_layout()
        x0, t, noise = _train_inputs(global_batch)
        a, b = shard_range(global_batch, rank, world)
        flat = _oracle_flat_grads(x0[a:b], t[a:b], noise[a:b], layout, n)
        scale = all_reduce_flat_grads(flat)
        if rank == 0:
            np.save(out_path, (flat * scale).numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_all_reduce_equals_full_batch_gradient(tmp_path):
    out = str(tmp_path / "grads.npy")
    mp.spawn(_train_worker, args=(2, _free_port(), 4, out), nprocs=2, join=True)
    layout, n = _train_layout()
    assert set(layout) == set(_train_shapes())
    want = _oracle_flat_grads(*_train_inputs(4), layout, n).numpy()
    got = np.load(out)
    assert np.abs(got - want).max() <= 1e-6 + 1e-4 * np.abs(want).max()


# ---- data-parallel training: replicas that start apart are brought together before the first exchange -------------------
def _sync_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diffusion_model_nemo_amd.parallel import all_reduce_flat_grads, broadcast_training_state
        gen = torch.Generator().manual_seed(100 + rank)          # every rank starts from its own initialisation
        params, m, v = (torch.randn(1000, generator=gen) for _ in range(3))
        step = broadcast_training_state((params, m, v), 7 + rank)
        assert step == 7
        grads = torch.randn(1000, generator=gen)                  # per-rank gradients
        scale = all_reduce_flat_grads(grads)
        params -= 0.1 * grads * scale                              # any deterministic update of identical inputs
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), torch.stack([params, m, v]).numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_replicas_hold_equal_parameters_after_a_step(tmp_path):
    mp.spawn(_sync_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b)
    want = torch.randn(1000, generator=torch.Generator().manual_seed(100))   # rank 0's parameters won
    g0 = torch.Generator().manual_seed(100)
    p0 = torch.randn(1000, generator=g0)
    assert torch.equal(p0, want)
    assert np.array_equal(a[1], torch.randn(1000, generator=g0).numpy())     # rank 0's first moment everywhere


# ---- data-parallel training: the broadcast comes BEFORE the first gradient is taken ------------------------------------
def _cpu_trainer(seed):
    """`UnetTrainer` with its device buffers replaced by CPU tensors and its gradient kernels by a closed form
    (grad = 2 * params * data): what is under test is the ORDER of its calls, which needs no GPU."""
    from diffusion_model_nemo_amd.training import UnetTrainer

    class CpuTrainer(UnetTrainer):
        def __init__(self):   # noqa: D401 - no native handle, no device
            gen = torch.Generator().manual_seed(seed)
            self.params, self.exp_avg, self.exp_avg_sq = (torch.randn(64, generator=gen) for _ in range(3))
            self.grads = torch.zeros(64)
            self.step_count = seed
            self._synced = False
            self._handle = None
            self.unet = type("Net", (), {"_weights_version": 1})()

        def loss_and_grads(self, data):
            self._before_step()                     # what the real loss_and_grads does in front of its launches
            self.grads = 2.0 * self.params * data
            return (self.params ** 2 * data).sum()

    return CpuTrainer()


def _first_step_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    tr = _cpu_trainer(100 + rank)                   # every rank starts from its own parameters
    data = torch.full((64,), float(rank + 1))
    tr.loss_and_grads(data)                         # no process group yet: nothing to synchronise with ...
    assert not tr._synced                           # ... and the trainer must not believe that it has
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tr.loss_and_grads(data)                     # the first gradient under the group is taken at rank 0's parameters
        assert tr._synced and tr.step_count == 100
        scale = tr.all_reduce_grads()
        np.save(os.path.join(out_dir, f"first_step_rank{rank}.npy"), torch.stack([tr.params, tr.grads * scale]).numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_replicas_are_synchronised_before_the_first_gradient(tmp_path):
    mp.spawn(_first_step_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "first_step_rank0.npy"), np.load(tmp_path / "first_step_rank1.npy")
    assert np.array_equal(a, b)
    p0 = _cpu_trainer(100).params                  # rank 0's start
    assert np.array_equal(a[0], p0.numpy())
    want = (2.0 * p0 * 1.0 + 2.0 * p0 * 2.0) / 2    # mean over the ranks of gradients taken at THE SAME parameters
    assert np.allclose(a[1], want.numpy(), rtol=1e-6, atol=0)


# ---- bench.py's multi-rank control flow, rehearsed on the CPU ------------------------------------------------------------
def _bench_worker(rank, world, port, out_dir, fail_rank):
    """What `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...` gives every rank: the rendezvous
    variables, then bench.main(). stdout / stderr go to one file per rank."""
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "LOCAL_RANK": str(rank),
                       "WORLD_SIZE": str(world)})
    if fail_rank is not None:
        os.environ["DMN_BENCH_DRY_FAIL_RANK"] = str(fail_rank)
    sys.stdout = open(os.path.join(out_dir, f"bench_rank{rank}.out"), "w")
    sys.stderr = open(os.path.join(out_dir, f"bench_rank{rank}.err"), "w")
    import bench
    bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--batch", "5", "--image-size", "8",
                "--backend", "gloo", "--dry-run"])
    assert not dist.is_initialized()          # the process group was destroyed on the way out
    sys.stdout.flush()
    sys.stderr.flush()


@pytest.mark.parametrize("fail_rank", [None, 1])
def test_bench_main_runs_two_ranks_through_its_control_flow(tmp_path, fail_rank):
    """Process-group set-up from the launcher's environment, image_offset = rank * B (checked inside bench.py through the
    index-valued images), the gather, the MIN-reduced agreement of the training extra (with one rank failing it: the
    entry is dropped on every rank and nothing hangs), MAX-reduced times, ONE JSON line on rank 0 only, clean exit."""
    import json
    mp.spawn(_bench_worker, args=(2, _free_port(), str(tmp_path), fail_rank), nprocs=2, join=True)
    lines0 = [ln for ln in open(tmp_path / "bench_rank0.out").read().splitlines() if ln.strip()]
    lines1 = [ln for ln in open(tmp_path / "bench_rank1.out").read().splitlines() if ln.strip()]
    assert len(lines0) == 1 and lines1 == [], (lines0, lines1)
    res = json.loads(lines0[0])
    assert res["n_gpus"] == 2 and res["steps"] == 3 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 10 and res["config"]["parallelism"] == "dp2"
    assert res["dry_run"] is True and res["backend"] == "gloo" and res["handoff_giveups"] == 0 and res["value"] > 0
    if fail_rank is None:
        assert res["train_step"]["n_gpus"] == 2 and res["train_step"]["config"]["global_batch"] == 64
    else:
        assert "train_step" not in res
        assert "injected failure" in open(tmp_path / f"bench_rank{fail_rank}.err").read()
