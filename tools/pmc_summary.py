#!/usr/bin/env python3
"""Per-kernel counter table of one denoising step from a rocprofv3 --pmc counter_collection CSV."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    k = int(r['Dispatch_Id'])
    e = d.setdefault(k, {'name': r['Kernel_Name'], 'grid': r['Grid_Size'], 't0': int(r['Start_Timestamp']), 't1': int(r['End_Timestamp'])})
    e[r['Counter_Name']] = e.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
ks = sorted(d)
tm = [k for k in ks if 'time_mlp' in d[k]['name'] or 'ingest_time' in d[k]['name']]
s, e = tm[1], tm[2]
cols = sys.argv[2:]
for k in ks:
    if k < s or k >= e: continue
    x = d[k]
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', x['name']).replace('(anonymous namespace)::', '').replace('void ', '')[:46]
    w = x.get('SQ_WAVES', 1) or 1
    print(f"{n:46s} g={x['grid']:>8s} us={(x['t1']-x['t0'])/1e3:6.1f} " + " ".join(f"{c.replace('SQ_','')}={x.get(c,0)/w:8.1f}" for c in cols))
