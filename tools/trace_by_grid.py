#!/usr/bin/env python3
"""Condense a rocprofv3 kernel-trace CSV into one line per (kernel, grid): calls, total and average duration.

usage: tools/trace_by_grid.py <kernel_trace.csv> > summary.csv
rocprofv3's own kernel_stats.csv averages over every launch of a kernel NAME; the convolution template serves
several tensor shapes under one name, so the per-grid average is what compares with a single-shape measurement.
"""
import collections
import csv
import sys

agg = collections.OrderedDict()
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        key = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(key, [0, 0])
        a[0] += 1
        a[1] += d
w = csv.writer(sys.stdout)
w.writerow(["Name", "GridX", "GridY", "GridZ", "Calls", "TotalDurationNs", "AverageNs"])
for (n, gx, gy, gz), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    w.writerow([n, gx, gy, gz, c, t, round(t / c, 1)])
