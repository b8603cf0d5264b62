#!/usr/bin/env python3
"""Counters per kernel (name x grid) from rocprofv3 --pmc runs: usage pmc_by_kernel.py <out.txt> <dir> [<dir> ...].
Every directory is one counter pass of the same command; counters are summed over a kernel's dispatches and printed per
wave (SQ counters count quad-cycles: x4 = cycles) beside the average duration."""
import collections
import csv
import glob
import os
import re
import sys

dst, dirs = sys.argv[1], sys.argv[2:]
agg = collections.OrderedDict()
names = []
for d in dirs:
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no counter_collection.csv under " + d
    disp = {}
    for f in files:
        for r in csv.DictReader(open(f)):
            k = (f, int(r["Dispatch_Id"]))
            e = disp.setdefault(k, {"name": r["Kernel_Name"], "grid": r["Grid_Size"],
                                    "us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "c": {}})
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for e in disp.values():
        n = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", e["name"])
        a = agg.setdefault((n, e["grid"]), {"calls": collections.defaultdict(float), "us": collections.defaultdict(float),
                                            "c": collections.defaultdict(float)})
        a["calls"][d] += 1
        a["us"][d] += e["us"]
        for c, v in e["c"].items():
            a["c"][c] += v
            if c not in names:
                names.append(c)
lines = ["kernel".ljust(64) + " grid".rjust(9) + " calls" + "  avg_us " + " ".join(c.replace("SQ_", "")[:14].rjust(14) for c in names)]
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]["us"].values()))
for (n, g), a in rows:
    d0 = dirs[0]
    calls = a["calls"].get(d0, 0) or max(a["calls"].values())
    us = (a["us"].get(d0, 0) or max(a["us"].values())) / calls
    waves = a["c"].get("SQ_WAVES", 0) / calls if a["c"].get("SQ_WAVES") else 0
    vals = []
    for c in names:
        v = a["c"].get(c, 0.0) / calls
        vals.append(f"{v / waves:14.1f}" if (waves and c.startswith("SQ_") and c != "SQ_WAVES") else f"{v:14.0f}")
    lines.append(f"{n[:64]:64s} {g:>9s} {int(calls):5d} {us:8.1f} " + " ".join(vals))
open(dst, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:40]))
