"""Sums FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, KiB) over the kernels of the LAST training step in a trace of
`train_bench.py --steps 1 --warmup 1` (two steps recorded: the second half of the dispatches), per kernel name.
FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes); WRITE_SIZE as reported.
usage: train_traffic.py <fetch dir> <write dir> <out.json>"""
import collections
import csv
import glob
import json
import sys


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = int(r["Dispatch_Id"])
        name = r["Kernel_Name"]
        per.setdefault(k, [name, 0.0])[1] += float(r["Counter_Value"])
    ids = sorted(per)
    # the two steps launch the same kernels: keep the second half
    half = ids[len(ids) // 2:]
    agg = collections.defaultdict(float)
    for k in half:
        n = per[k][0]
        n = n[n.find("::") + 2:] if "::" in n else n
        agg[n.split("(")[0]] += per[k][1] * 1024.0
    return agg


fetch = load(sys.argv[1], "FETCH_SIZE")
write = load(sys.argv[2], "WRITE_SIZE")
rows = []
for n in sorted(set(fetch) | set(write), key=lambda n: -(2 * fetch.get(n, 0) + write.get(n, 0))):
    rows.append({"kernel": n, "read_GB": 2 * fetch.get(n, 0) / 1e9, "write_GB": write.get(n, 0) / 1e9})
tot_r, tot_w = sum(r["read_GB"] for r in rows), sum(r["write_GB"] for r in rows)
res = {"read_GB": tot_r, "write_GB": tot_w, "total_GB": tot_r + tot_w, "per_kernel": rows[:20],
       "correction": "FETCH_SIZE x2 on gfx950; WRITE_SIZE as reported"}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "per_kernel"}))
for r in rows[:14]:
    print(f"{r['kernel'][:70]:70s} read {r['read_GB']:7.3f} GB  write {r['write_GB']:7.3f} GB")
