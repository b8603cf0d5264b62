#!/usr/bin/env python3
"""Print the per-kernel timeline of one denoising step from a rocprofv3 kernel-trace CSV."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'time_mlp' in r['Kernel_Name'] or 'ingest_time' in r['Kernel_Name']]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
s, e = idx[k], idx[k + 1]
t0 = int(rows[s]['Start_Timestamp'])
agg = {}
for r in rows[s:e]:
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', r['Kernel_Name']).replace('(anonymous namespace)::', '')
    n = re.sub(r'^void ', '', n)[:58]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if '-q' not in sys.argv:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} {d:7.1f}us grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} lds={r['LDS_Block_Size']} vgpr={r['VGPR_Count']} {n}")
    key = n.split('(')[0]
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += d
print("---- per kernel (one step)")
tot = 0
for k2, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:8.1f}us {c:3d}x {k2}"); tot += d
print(f"{tot:8.1f}us total busy; span {(int(rows[e]['Start_Timestamp'])-t0)/1e3:.1f}us")
