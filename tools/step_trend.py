"""Step duration against step number inside one sampling loop, from a rocprofv3 kernel-trace CSV: the distance between the
starts of consecutive opening kernels (ingest_time_kernel) per launch chain, averaged over blocks of 50 steps.
usage: python tools/step_trend.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

starts = defaultdict(list)
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        if "ingest_time_kernel" in r["Kernel_Name"]:
            starts[(r.get("Queue_Id"), r.get("Grid_Size_Y", r.get("Grid_Size")))].append(int(r["Start_Timestamp"]))
for key, ts in sorted(starts.items()):
    ts.sort()
    # loops are separated by gaps far longer than a step
    loops, cur = [], [ts[0]]
    for a, b in zip(ts, ts[1:]):
        if b - a > 5_000_000:
            loops.append(cur)
            cur = []
        cur.append(b)
    loops.append(cur)
    for li, lp in enumerate(loops):
        if len(lp) < 200:
            continue
        d = [(b - a) / 1e3 for a, b in zip(lp, lp[1:])]
        blocks = [sum(d[i:i + 50]) / len(d[i:i + 50]) for i in range(0, len(d), 50)]
        print(f"TREND queue/grid {key} loop {li} ({len(lp)} steps): " + " ".join(f"{b:.0f}" for b in blocks), flush=True)
