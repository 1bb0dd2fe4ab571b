#!/usr/bin/env python3
"""Parse a rocprofv3 --pmc counter CSV of tools/ablate.py: per-variant mean of a counter over the timed step dispatches
(each variant = 5 warm-up + k timed launches of k_env<..., true>).

Collect it like this (GPU box): the step count is exported in the CALLING shell and python3 comes directly after `--` -- on
this pool nothing may sit between the profiler and the interpreter (no `env VAR=...`, no shebang hop), because the
profiler's preloaded library has already initialised the GPU and every such hop is an exec:
    cd /tmp && export TMPDIR=/tmp ABLATE_STEPS=10
    rocprofv3 --pmc SQ_INSTS_VALU -d out -o p --output-format csv -- python3 $REPO/tools/ablate.py
    python3 $REPO/tools/ablate_pmc.py out/.../p_counter_collection.csv 10
"""
import collections
import csv
import sys

path, k = sys.argv[1], int(sys.argv[2])
names = ["full", "forces+prior+integrate", "pair masks", "cell scan", "occupied filter", "rank-select bits", "reward sums",
         "obs head pairs", "obs sensed pairs", "cell staging", "ordered insertion", "emit walk", "nearest merge"]
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
    tot = 0.0
    for i, nm in enumerate(names):
        chunk = vals[i * per + 5:(i + 1) * per]
        if not chunk:
            break
        m = sum(chunk) / len(chunk)
        if base is None:
            base = m
        ph = (m - base) / 2 / 4096
        tot = tot + ph if i else 0.0
        print(f"  {nm:24s} {m:14.4g}   per env {m / 4096:9.1f}   phase per env {ph:9.1f}")
    print(f"  sum of phases per env {tot:9.1f}   remainder {base / 4096 - tot:9.1f}")
