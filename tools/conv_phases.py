"""Tuning aid (GPU): per-workgroup phase timestamps of the convolution kernel on the network's main shapes.

usage: python tools/conv_phases.py [B]
Prints, per shape, the median cycles spent in: coefficient set-up, tile staging, MFMA loop, epilogue, and the
wall-clock span of the launch (first workgroup start to last workgroup end).
"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402


def run(name, B, H, W, cin, cout, k, stride, pad, transposed, pro, epi_G, dtype=1, tile=0, bn=0, c1=0):
    lib = N.lib()
    g = torch.Generator().manual_seed(1)
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    x0 = torch.randn((B, H, W, cin - c1), generator=g).to(tdt).cuda()
    x1 = torch.randn((B, H, W, c1), generator=g).to(tdt).cuda() if c1 else None
    wshape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    w = (torch.randn(wshape, generator=g) * (1.0 / (k * k * cin)) ** 0.5).numpy().astype(np.float32)
    b = torch.randn((cout,), generator=g).numpy().astype(np.float32)
    Ho, Wo = (2 * H, 2 * W) if transposed else ((H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1)
    out = torch.zeros((B, Ho, Wo, cout), device="cuda", dtype=tdt)
    pg = pb = pst = None
    pG = 0
    if pro:
        pG = 8
        pg = np.ones(cin, np.float32)
        pb = np.zeros(cin, np.float32)
        st = torch.zeros((B, pG, 1, 4))   # one partial per group {sum 0, squared deviations = count, count, -}: mean 0, variance 1
        st[..., 1] = float(H * W * cin // pG)
        st[..., 2] = float(H * W * cin // pG)
        pst = st.cuda()
    est = torch.zeros((B * max(epi_G, 1) * 4096 * 4,), device="cuda") if epi_G else None
    dbg = torch.zeros((1 << 16, 8), dtype=torch.int64, device="cuda")
    slots = C.c_int(0)
    N.check(lib.dmn_test_set_conv_debug(dbg.data_ptr()))
    for _ in range(2):
        dbg.zero_()
        torch.cuda.synchronize()
        N.check(lib.dmn_test_conv2d(
            dtype, x0.data_ptr(), None if x1 is None else x1.data_ptr(), cin - c1, c1, B, H, W, w.ctypes.data, b.ctypes.data,
            cout, k, k, stride, pad, 1 if transposed else 0, out.data_ptr(), None, None if pg is None else pg.ctypes.data,
            None if pb is None else pb.ctypes.data, None, None if pst is None else pst.data_ptr(), pG, 1, 1,
            None if est is None else est.data_ptr(), epi_G, C.byref(slots), tile, bn, None))
        torch.cuda.synchronize()
    N.check(lib.dmn_test_set_conv_debug(None))
    d = dbg.cpu().numpy()
    d = d[d[:, 0] != 0]
    nwg = len(d)
    ph = np.stack([d[:, 1] - d[:, 7], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 4] - d[:, 3], d[:, 4] - d[:, 7]], 1)
    wall = (d[:, 5].max() - d[:, 0].min()) / 100.0   # 100 MHz counter -> us
    life = (d[:, 5] - d[:, 0]) / 100.0
    cyc_per_us = np.median((d[:, 4] - d[:, 7]) / np.maximum(life, 0.01))
    med = np.median(ph, 0)
    st0 = (d[:, 0] - d[:, 0].min()) / 100.0
    starts = np.sort(st0)
    if "-s" in sys.argv:   # start-time distribution: workgroups started by 0.5, 1, 1.5, 2, 3, 4, 6, 8, 12 us
        print("    started by us:", " ".join(f"{t}:{int((st0 <= t).sum())}" for t in (0.5, 1, 1.5, 2, 3, 4, 6, 8, 12)),
              " ends p50/p90/max:", " ".join(f"{np.percentile((d[:, 5] - d[:, 0].min()) / 100.0, q):.1f}" for q in (50, 90, 100)))
    if starts[-1] > 3.0:   # more than one residency round: medians of the first and of the later workgroups
        for nm, sel in (("first round", st0 < 1.5), ("later rounds", st0 >= 1.5)):
            m = np.median(ph[sel], 0)
            print(f"    {nm:13s} wgs={int(sel.sum()):5d} coef={m[0]:6.0f} stage={m[1]:6.0f} mfma={m[2]:6.0f} epi={m[3]:6.0f} "
                  f"total={m[4]:6.0f}  life={np.median(life[sel]):5.1f}us")
    print(f"{name:34s} wgs={nwg:5d} span={wall:6.1f}us  wg life med={np.median(life):5.1f}us  cycles: coef={med[0]:6.0f} "
          f"stage={med[1]:6.0f} mfma={med[2]:6.0f} epi={med[3]:6.0f} total={med[4]:6.0f}  (~{cyc_per_us:.0f} cyc/us) "
          f"last start={starts[-1]:.1f}us")


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "-s" else 64
    if len(sys.argv) > 2 and sys.argv[2] == "cfg":
        for tile, bn in ((0, 0), (16, 32), (16, 64), (8, 32), (8, 64)):
            print(f"--- tile {tile} bn {bn}")
            run("L1 3x3 64->64 32x32 pro+stats", B, 32, 32, 64, 64, 3, 1, 1, False, True, 8, tile=tile, bn=bn)
            run("L1 3x3 128->64 32x32 concat", B, 32, 32, 128, 64, 3, 1, 1, False, False, 8, c1=64, tile=tile, bn=bn)
            run("L2 3x3 128->128 16x16 pro+stats", B, 16, 16, 128, 128, 3, 1, 1, False, True, 8, tile=tile, bn=bn)
            run("L2 3x3 256->128 16x16 concat", B, 16, 16, 256, 128, 3, 1, 1, False, False, 8, c1=128, tile=tile, bn=bn)
            run("L3 3x3 256->256 8x8 pro+stats", B, 8, 8, 256, 256, 3, 1, 1, False, True, 8, tile=tile, bn=bn)
            if tile != 16:
                run("L1 4x4s2 64->64 32x32", B, 32, 32, 64, 64, 4, 2, 1, False, False, 0, tile=tile, bn=bn)
            run("L1 convT 64->64 32x32", B, 32, 32, 64, 64, 4, 2, 1, True, False, 0, tile=tile, bn=bn)
            run("L2 convT 128->128 16x16", B, 16, 16, 128, 128, 4, 2, 1, True, False, 0, tile=tile, bn=bn)
        sys.exit(0)
    run("L0 3x3 32->32 64x64 pro+stats", B, 64, 64, 32, 32, 3, 1, 1, False, True, 8)
    run("L0 3x3 32->32 64x64 plain", B, 64, 64, 32, 32, 3, 1, 1, False, False, 0)
    run("L0 3x3 64->32 64x64 concat", B, 64, 64, 64, 32, 3, 1, 1, False, False, 8, c1=32)
    run("L1 3x3 64->64 32x32 pro+stats", B, 32, 32, 64, 64, 3, 1, 1, False, True, 8)
    run("L2 3x3 128->128 16x16 pro+stats", B, 16, 16, 128, 128, 3, 1, 1, False, True, 8)
    run("L3 3x3 256->256 8x8 pro+stats", B, 8, 8, 256, 256, 3, 1, 1, False, True, 8)
    run("L3 3x3 512->256 8x8 concat", B, 8, 8, 512, 256, 3, 1, 1, False, False, 8, c1=256)
    run("L0 4x4s2 32->32 64x64", B, 64, 64, 32, 32, 4, 2, 1, False, False, 0)
    run("L1 convT 64->64 32x32", B, 32, 32, 64, 64, 4, 2, 1, True, False, 0)
    run("L0 1x1 64->32 64x64", B, 64, 64, 64, 32, 1, 1, 0, False, False, 0, c1=32)
