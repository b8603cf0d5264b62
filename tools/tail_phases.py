"""Tuning aid (GPU): block2.proj + block tail in one launch (in-launch statistics hand-off) against the two-launch form.

usage: python tools/tail_phases.py [B]
Per shape: bit-equality of the two forms' outputs, and per-workgroup phase times of the fused launch (set-up, staging,
MFMA loop, statistics + wait, apply) with the launch's span; the two-launch form's spans beside it.
"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from diffusion_model_nemo_amd import _native as N  # noqa: E402


def run(name, B, H, C_, G, dtype=1, out_G=1, plan_B=64, tall=0):
    lib = N.lib()
    g = torch.Generator().manual_seed(1)
    tdt = torch.bfloat16 if dtype == 1 else torch.float32
    x = torch.randn((B, H, H, C_), generator=g).to(tdt).cuda()
    res = torch.randn((B, H, H, C_), generator=g).to(tdt).cuda()
    w = (torch.randn((C_, C_, 3, 3), generator=g) * (1.0 / (9 * C_)) ** 0.5).numpy().astype(np.float32)
    bias = torch.randn((C_,), generator=g).numpy().astype(np.float32)
    pg, pb = np.ones(C_, np.float32), np.zeros(C_, np.float32)
    tg = (1.0 + 0.1 * torch.randn((C_,), generator=g)).numpy().astype(np.float32)
    tb = (0.1 * torch.randn((C_,), generator=g)).numpy().astype(np.float32)
    st = torch.zeros((B, G, 1, 4))   # one partial per group {sum 0, squared deviations = count, count, -}: mean 0, variance 1
    st[..., 1] = float(H * H * C_ // G)
    st[..., 2] = float(H * H * C_ // G)
    pst = st.cuda()
    outs, stats, spans = [], [], []
    dbg = torch.zeros((1 << 16, 8), dtype=torch.int64, device="cuda")
    for fused in (1 + tall, tall):   # tall = 2: 32x16-pixel tiles in both forms
        out = torch.zeros((B, H, H, C_), device="cuda", dtype=tdt)
        h2 = torch.zeros((B, H, H, C_), device="cuda", dtype=tdt)
        ost = torch.zeros((B * out_G * 4096 * 4,), device="cuda")
        slots, give = C.c_int(0), C.c_uint(0)
        N.check(lib.dmn_test_set_conv_debug(dbg.data_ptr()))
        if fused & 1:   # shapes outside the exchange's limits (workgroups per image, LDS per workgroup) are refused
            rc = lib.dmn_test_conv_block_tail(
                dtype, x.data_ptr(), C_, B, H, H, w.ctypes.data, bias.ctypes.data, C_, pg.ctypes.data, pb.ctypes.data,
                pst.data_ptr(), G, 1, tg.ctypes.data, tb.ctypes.data, G, res.data_ptr(), h2.data_ptr(), out.data_ptr(),
                ost.data_ptr(), out_G, C.byref(slots), fused, plan_B, 0, 0, C.byref(give), None)
            if rc == -5:
                N.check(lib.dmn_test_set_conv_debug(None))
                print(f"{name:28s} refused: outside the in-launch exchange's limits (kept on the two-launch form)", flush=True)
                return
        for _ in range(3):
            dbg.zero_()
            torch.cuda.synchronize()
            N.check(lib.dmn_test_conv_block_tail(
                dtype, x.data_ptr(), C_, B, H, H, w.ctypes.data, bias.ctypes.data, C_, pg.ctypes.data, pb.ctypes.data,
                pst.data_ptr(), G, 1, tg.ctypes.data, tb.ctypes.data, G, res.data_ptr(), h2.data_ptr(), out.data_ptr(),
                ost.data_ptr(), out_G, C.byref(slots), fused, plan_B, 0, 0, C.byref(give), None))
            torch.cuda.synchronize()
        N.check(lib.dmn_test_set_conv_debug(None))
        assert give.value == 0, give.value
        d = dbg.cpu().numpy()
        d = d[d[:, 0] != 0]
        span = (d[:, 5].max() - d[:, 0].min()) / 100.0
        if fused & 1:
            ph = np.stack([d[:, 1] - d[:, 7], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2], d[:, 6] - d[:, 3], d[:, 4] - d[:, 6]], 1)
            med = np.median(ph, 0)
            life = np.median((d[:, 5] - d[:, 0]) / 100.0)
            last = ((d[:, 0] - d[:, 0].min()) / 100.0).max()
            txt = (f"wgs={len(d):5d} span={span:5.1f}us life={life:5.1f}us last start={last:4.1f}us cycles: setup={med[0]:5.0f} "
                   f"stage={med[1]:5.0f} mfma={med[2]:5.0f} stats+wait={med[3]:5.0f} apply={med[4]:5.0f}")
        spans.append(span)
        outs.append(out)
        n = slots.value
        stats.append(ost[:B * out_G * n * 4].view(B, out_G, n, 4)[..., 0].sum(2).cpu())   # group sums (slot 0 of every partial)
    same = torch.equal(outs[0], outs[1])
    sd = float((stats[0] - stats[1]).abs().max() / stats[1].abs().max())
    print(f"{name:28s} out bit-equal {same}  out-stats rel diff {sd:.1e}  fused {txt}  | conv alone span {spans[1]:5.1f}us", flush=True)
    assert same and sd < 1e-4


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for dt in (1, 0):
        print("bf16" if dt == 1 else "fp32")
        run("L1 64->64 32x32", B, 32, 64, 8, dt)
        run("L2 128->128 16x16", B, 16, 128, 8, dt)
        run("L3 256->256 8x8", B, 8, 256, 8, dt)
        run("L3 256->256 8x8 groups 4", B, 8, 256, 4, dt)
        run("ragged 64->64 24x24", 3, 24, 64, 8, dt)
        run("decoder 32->32 32x32 (G_out 8)", B, 32, 32, 8, dt, out_G=8)
        run("L0 32->32 64x64, 32x16 tiles", min(B, 32), 64, 32, 8, dt, tall=2)
        run("final 32->32 64x64 (G_out 8), 32x16", min(B, 32), 64, 32, 8, dt, out_G=8, tall=2)
