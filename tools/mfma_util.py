#!/usr/bin/env python3
"""Matrix-core utilisation per kernel (name x grid) from one rocprofv3 counter pass.

usage: mfma_util.py <rocprof output dir> <summary.txt>
SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of a kernel is divided by the same quotient of the bare
v_mfma_f32_16x16x32_bf16 loop (mfma_loop_kernel<0>, four waves per SIMD), which keeps every matrix unit busy:
the result is the fraction of the launch during which the matrix units worked.
"""
import collections
import csv
import glob
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
assert files, "no counter_collection.csv under " + src
disp = {}
for f in files:
    for r in csv.DictReader(open(f)):
        k = (f, int(r["Dispatch_Id"]))
        e = disp.setdefault(k, {"name": r["Kernel_Name"], "grid": r["Grid_Size"],
                                "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.OrderedDict()
for e in disp.values():
    n = re.sub(r"\(anonymous namespace\)::|void ", "", e["name"])
    a = agg.setdefault((n, e["grid"]), collections.defaultdict(float))
    a["calls"] += 1
    for c in ("us", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "GRBM_GUI_ACTIVE"):
        a[c] += e.get(c, 0.0)
cal = [a for (n, g), a in agg.items() if n.startswith("mfma_loop_kernel<0>") and a["GRBM_GUI_ACTIVE"] > 0]
cal = max((a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["GRBM_GUI_ACTIVE"] for a in cal), default=None)
lines = [f"calibration: bare bf16 MFMA loop MFMA_BUSY/GUI_ACTIVE = {cal}"]
lines.append(f"{'kernel':90s} {'grid':>9s} {'calls':>5s} {'avg us':>8s} {'mfma busy/active':>17s} {'utilisation':>11s}")
rows = sorted(agg.items(), key=lambda kv: -kv[1]["us"])
for (n, g), a in rows:
    if a["GRBM_GUI_ACTIVE"] <= 0:
        continue
    q = a["SQ_VALU_MFMA_BUSY_CYCLES"] / a["GRBM_GUI_ACTIVE"]
    u = f"{q / cal:10.1%}" if cal else "         -"
    lines.append(f"{n[:90]:90s} {g:>9s} {int(a['calls']):5d} {a['us'] / a['calls']:8.1f} {q:17.3f} {u}")
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:45]))
