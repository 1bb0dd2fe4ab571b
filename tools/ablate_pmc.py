#!/usr/bin/env python3
"""Parse a rocprofv3 --pmc counter CSV of `ABLATE_STEPS=k python tools/ablate.py`: per-variant mean of a counter
over the timed step dispatches (each variant = 5 warm-up + k timed launches of k_env<..., true>)."""
import collections
import csv
import sys

path, k = sys.argv[1], int(sys.argv[2])
names = ["full", "forces+prior x3", "neighbour search x3", "cell scan x3", "occupied filter x3", "list emit x3",
         "reward sums x3", "obs head x3", "obs sensed x3"]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if "k_env<" in r["Kernel_Name"] and "true>" in r["Kernel_Name"]:
        rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for cname, v in rows.items():
    v.sort()
    vals = [x[1] for x in v]
    vals = vals[100:]                       # the 100 assembling steps of the set-up phase
    per = 5 + k
    print(cname)
    base = None
    for i, nm in enumerate(names):
        chunk = vals[i * per + 5:(i + 1) * per]
        if not chunk:
            break
        m = sum(chunk) / len(chunk)
        if base is None:
            base = m
        print(f"  {nm:22s} {m:14.4g}   per env {m / 4096:9.1f}   phase per env {(m - base) / 2 / 4096:9.1f}")
