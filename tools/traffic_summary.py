#!/usr/bin/env python3
"""HBM traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: tools/traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel substring>
           <grid size> <out.json> [batch image_size dtype]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB. On gfx950 FETCH_SIZE counts 64 B per 128-B request of
wide coalesced reads, i.e. half the bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.
"""
import csv
import json
import sys


def mean_counter(path, name_sub, grid, counter):
    per_dispatch = {}
    for r in csv.DictReader(open(path)):
        if name_sub in r["Kernel_Name"] and r["Counter_Name"] == counter and int(r["Grid_Size"]) == grid:
            per_dispatch[r["Dispatch_Id"]] = per_dispatch.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    v = list(per_dispatch.values())
    return sum(v) / len(v), len(v)


fetch_csv, write_csv, sub, grid, out = sys.argv[1:6]
grid = int(grid)
f_kib, nf = mean_counter(fetch_csv, sub, grid, "FETCH_SIZE")
w_kib, nw = mean_counter(write_csv, sub, grid, "WRITE_SIZE")
res = {
    "kernel": sub, "grid_size": grid, "launches_averaged": {"FETCH_SIZE": nf, "WRITE_SIZE": nw},
    "FETCH_SIZE_KiB_raw": f_kib, "WRITE_SIZE_KiB_raw": w_kib,
    "read_bytes_corrected": 2.0 * f_kib * 1024.0, "write_bytes": w_kib * 1024.0,
    "hbm_bytes_per_launch": 2.0 * f_kib * 1024.0 + w_kib * 1024.0,
    "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported",
}
if len(sys.argv) > 8:
    res.update(batch=int(sys.argv[6]), image_size=int(sys.argv[7]), dtype=sys.argv[8])
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
