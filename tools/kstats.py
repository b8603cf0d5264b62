"""Prints the top rows of a rocprofv3 kernel_stats.csv as ms per step: python tools/kstats.py <dir> <steps>"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 20]:
    print(f"{r['Name'][:80]:80s} calls={r['Calls']:>5s} {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step "
          f"avg={float(r['AverageNs']) / 1e3:8.1f} us {float(r['TotalDurationNs']) / tot * 100:5.1f}%")
print("total ms/step", tot / 1e6 / steps)
