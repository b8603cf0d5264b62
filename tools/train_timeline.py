"""Concurrency picture of ONE training step from a rocprofv3 --kernel-trace CSV: per hardware queue the busy time and
span, how long 0/1/2/3+ kernels were running at once, and what ran in the step's last 600 us.
usage: train_timeline.py <rocprof dir> [out.txt]"""
import collections
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["t0"], r["t1"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    r["n"] = re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"])[:60]
rows.sort(key=lambda r: r["t0"])
rep = [i for i, r in enumerate(rows) if r["n"].startswith("repack_kernel")]
adam = [i for i, r in enumerate(rows) if r["n"].startswith("adamw_kernel")]
a = rep[-1]
b = [i for i in adam if i > a][0]
step = rows[a:b + 1]
T0, T1 = step[0]["t0"], max(r["t1"] for r in step)
out = [f"step: {len(step)} launches, {(T1 - T0) / 1e3:.1f} us from repack start to adamw end"]
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
byq = collections.defaultdict(list)
for r in step:
    byq[r[qkey]].append(r)
for q, rs in byq.items():
    busy = sum(r["t1"] - r["t0"] for r in rs)
    out.append(f"queue {q}: {len(rs):4d} launches, busy {busy / 1e3:8.1f} us, first {(rs[0]['t0'] - T0) / 1e3:8.1f}, last end "
               f"{(max(r['t1'] for r in rs) - T0) / 1e3:8.1f}; top: " +
               ", ".join(f"{n[:28]} {v / 1e3:.0f}" for n, v in collections.Counter(
                   {k: sum(r['t1'] - r['t0'] for r in rs if r['n'] == k) for k in {r['n'] for r in rs}}).most_common(3)))
ev = sorted([(r["t0"], 1) for r in step] + [(r["t1"], -1) for r in step])
lvl, last, hist = 0, T0, collections.Counter()
for t, d in ev:
    hist[min(lvl, 4)] += t - last
    last = t
    lvl += d
out.append("kernels in flight -> us: " + ", ".join(f"{k}: {v / 1e3:.0f}" for k, v in sorted(hist.items())))
out.append("last 600 us of the step:")
for r in step:
    if r["t1"] > T1 - 600000:
        out.append(f"  q{r[qkey]:>3s} {(r['t0'] - T0) / 1e3:9.1f} +{(r['t1'] - r['t0']) / 1e3:7.1f}  {r['n']}  grid {r.get('Grid_Size', '')}")
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
