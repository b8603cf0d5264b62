"""Tuning aid (GPU): per-workgroup phase timestamps of the two fused linear-attention kernels on the network's shapes.

usage: python tools/la_phases.py [B]
"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402


def summarize(tag, d, names):
    d = d[d[:, 0] != 0]
    if len(d) == 0:
        print(tag, "no records")
        return
    ph = np.stack([d[:, 1] - d[:, 7], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 4] - d[:, 3], d[:, 4] - d[:, 7]], 1)
    wall = (d[:, 5].max() - d[:, 0].min()) / 100.0
    life = (d[:, 5] - d[:, 0]) / 100.0
    med = np.median(ph, 0)
    starts = (d[:, 0] - d[:, 0].min()) / 100.0
    print(f"  {tag:4s} wgs={len(d):5d} span={wall:6.1f}us life med={np.median(life):5.1f}us cycles: "
          + " ".join(f"{n}={v:6.0f}" for n, v in zip(names, med)) + f" last start={starts.max():.1f}us")


def run(B, n, Cch, dtype=1):
    lib = N.lib()
    g = torch.Generator().manual_seed(3)
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    x = torch.randn((B, n, Cch), generator=g)
    s1 = x.sum((1, 2))
    cnt = torch.full_like(s1, float(n * Cch))
    m2 = ((x - (s1 / cnt)[:, None, None]) ** 2).sum((1, 2))
    xs = torch.stack([s1, m2, cnt, torch.zeros_like(s1)], 1).contiguous().cuda()   # one partial per image {sum, squared deviations, count, -}
    xd = x.to(tdt).cuda()
    wqkv = (torch.randn((384, Cch), generator=g) * Cch ** -0.5).numpy().astype(np.float32)
    wout = (torch.randn((Cch, 128), generator=g) * 128 ** -0.5).numpy().astype(np.float32)
    bout = torch.randn((Cch,), generator=g).numpy().astype(np.float32)
    gamma = np.ones(Cch, np.float32)
    beta = np.zeros(Cch, np.float32)
    y = torch.zeros((B, n, Cch), device="cuda", dtype=tdt)
    ystats = torch.zeros((B, 1024, 4), device="cuda")
    scratch = torch.zeros((B * 4 * ((n + 31) // 32) * 1088,), device="cuda")
    dbg = torch.zeros((2 * 65536, 8), dtype=torch.int64, device="cuda")
    slices = C.c_int(0)
    N.check(lib.dmn_test_set_conv_debug(dbg.data_ptr()))
    for _ in range(2):
        dbg.zero_()
        torch.cuda.synchronize()
        N.check(lib.dmn_test_linattn(dtype, xd.data_ptr(), xs.data_ptr(), B, n, Cch, gamma.ctypes.data, beta.ctypes.data,
                                     wqkv.ctypes.data, wout.ctypes.data, bout.ctypes.data, y.data_ptr(), ystats.data_ptr(),
                                     C.byref(slices), scratch.data_ptr(), scratch.numel() * 4, None))
        torch.cuda.synchronize()
    N.check(lib.dmn_test_set_conv_debug(None))
    d = dbg.cpu().numpy()
    print(f"n={n} C={Cch} B={B}")
    summarize("ctx", d[:65536], ["coef", "stage", "sweep1", "sweep2", "total"])
    summarize("out", d[65536:], ["coef+stage", "combine", "tiles", "stats", "total"])


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for n, c in ((4096, 32), (1024, 64), (256, 128), (64, 256)):
        run(B, n, c)
