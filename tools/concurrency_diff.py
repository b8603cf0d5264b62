"""What differs in the workspace behind launch K of the network forward when two handles run at the same time, compared
with the same (truncated) forward run alone. Arguments: K, repetitions, dtype."""
import os
import re
import sys
import threading

import torch

sys.path.insert(0, ".")
K = int(sys.argv[1])
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
DTYPE = sys.argv[3] if len(sys.argv) > 3 else "bf16"
FILL = int(sys.argv[4]) if len(sys.argv) > 4 else 0   # byte the workspace is filled with before every forward (255: NaN patterns)
from diffusion_model_nemo_amd import _native as N  # noqa: E402
if os.environ.get("DMN_LIB_VARIANT"):
    N.LIB_PATH = N.LIB_PATH.replace("libdmn_hip.so", f"libdmn_hip_{os.environ['DMN_LIB_VARIANT']}.so")
from diffusion_model_nemo_amd.modules import Unet  # noqa: E402

dev = torch.device("cuda", 0)
SIZE, MULTS, B, SLOTS = 64, (1, 2, 4), 32, 256


def build():
    torch.manual_seed(1234)
    return Unet(input_dim=SIZE, dim=32, dim_mults=list(MULTS), channels=3, resnet_block_groups=8, use_convnext=False,
                dropout=0.0, compute_dtype=DTYPE).to(dev).eval()


lib = N.lib()
nets = [build() for _ in range(2)]
g = torch.Generator(device=dev).manual_seed(3)
xs = [torch.randn(B, 3, SIZE, SIZE, device=dev, generator=g) for _ in range(2)]
t = torch.full((B,), 500, device=dev, dtype=torch.long)
slots = [torch.zeros(SLOTS, dtype=torch.int64, device=dev) for _ in range(2)]
ref_ws, ref_sum, name = [], [], None
for i in range(2):
    nets[i](xs[i], t)
    h = nets[i].native_handle()
    N.check(lib.dmn_test_unet_trace(h, slots[i].data_ptr(), SLOTS))
    N.check(lib.dmn_test_unet_trace_stop(h, K))
    for _ in range(2):
        nets[i]._workspace.fill_(FILL)
        nets[i](xs[i], t)
        torch.cuda.synchronize()
    ref_ws.append(nets[i]._workspace.clone())
    ref_sum.append(slots[i][:K + 1].clone())
    name = lib.dmn_test_unet_trace_name(h, K).decode()
print(f"launch {K}: {name}", flush=True)

streams = [torch.cuda.Stream(dev) for _ in range(2)]
kept = [[], []]
nbad = [0, 0]


def work(i):
    with torch.cuda.stream(streams[i]):
        for _ in range(REPS):
            nets[i]._workspace.fill_(FILL)
            nets[i](xs[i], t)
            if not torch.equal(slots[i][:K + 1], ref_sum[i]):
                nbad[i] += 1
                if len(kept[i]) < 3:
                    kept[i].append(nets[i]._workspace.clone())


th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
for x in th:
    x.start()
for x in th:
    x.join()
torch.cuda.synchronize()
offs = {m.group(1): int(m.group(2)) for m in re.finditer(r"(\w+)@(\d+)", name)}
print("regions:", offs)
for i in range(2):
    print(f"handle {i}: {nbad[i]} of {REPS} runs differ behind launch {K}")
    for ws in kept[i]:
        d = (ws != ref_ws[i]).nonzero().flatten().cpu()
        if d.numel() == 0:
            print("   (workspace equal: only an earlier checksum differed)")
            continue
        # contiguous runs of differing bytes, merged when closer than 64 bytes
        runs = []
        start = prev = int(d[0])
        for v in d[1:].tolist():
            if v - prev > 64:
                runs.append((start, prev))
                start = v
            prev = v
        runs.append((start, prev))
        print(f"   {d.numel()} bytes differ in {len(runs)} runs; first runs: {runs[:12]}")
        if DTYPE == "bf16" and "y" in offs:
            yv = ws[offs["y"]:offs["stats"]].view(torch.bfloat16).float()
            print(f"      non-finite values in y: {int((~torch.isfinite(yv)).sum())} of {yv.numel()}")
        for nm, o in offs.items():
            inside = [(a - o, b - o) for a, b in runs if a >= o]
            if inside:
                print(f"      relative to {nm}: {inside[:8]}")
        if "stats" in offs:   # statistics are float32 pairs (sum, sum of squares) per slot
            o = offs["stats"]
            idx = sorted({(a - o) // 4 for a, _ in runs if a >= o})[:10]
            gf = ws[o:].view(torch.float32)
            ef = ref_ws[i][o:].view(torch.float32)
            recompute = None
            m = re.search(r"conv kind 0 (\d+)->(\d+) at (\d+)x(\d+)", name)
            if m and "out" in offs and DTYPE == "bf16":   # 16x16 tiles, 8 groups: statistics recomputed from the stored output
                co, hh = int(m.group(2)), int(m.group(3))
                outt = ref_ws[i][offs["out"]:offs["out"] + B * hh * hh * co * 2].view(torch.bfloat16).float().view(B, hh // 16, 16, hh // 16, 16, 8, co // 8)
                sm = outt.sum(dim=(2, 4, 6)).permute(0, 3, 1, 2).reshape(-1)        # [b][g][ty][tx]
                sq = (outt * outt).sum(dim=(2, 4, 6)).permute(0, 3, 1, 2).reshape(-1)
                recompute = (sm, sq)
            for j in idx:
                if recompute is not None and j // 2 < recompute[0].numel():
                    print(f"      from the stored output: {float(recompute[j & 1][j // 2]):.8g}")
                if j < gf.numel():
                    print(f"      stats float {j} (slot {j // 2}, {'sumsq' if j & 1 else 'sum'}): got {float(gf[j]):.8g} expected {float(ef[j]):.8g}"
                          f"   the pair's other half: got {float(gf[j ^ 1]):.8g} expected {float(ef[j ^ 1]):.8g}")
        a, b = runs[0]
        a &= ~1
        n = min(16, (b - a) // 2 + 1)
        got = ws[a:a + 2 * n].view(torch.bfloat16 if DTYPE == "bf16" else torch.float32).float().cpu().tolist()
        exp = ref_ws[i][a:a + 2 * n].view(torch.bfloat16 if DTYPE == "bf16" else torch.float32).float().cpu().tolist()
        print("      got     ", [f"{v:.5g}" for v in got])
        print("      expected", [f"{v:.5g}" for v in exp])
