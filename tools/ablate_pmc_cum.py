#!/usr/bin/env python3
"""Parse a rocprofv3 --pmc CSV of `tools/ablate.py --cumulative`: counters up to each exit point.

Collect (GPU box; variable exported in the calling shell, python3 directly after `--`, see tools/ablate_pmc.py):
    cd /tmp && export TMPDIR=/tmp ABLATE_STEPS=10
    rocprofv3 --pmc SQ_INSTS_VALU -d out -o p --output-format csv -- python3 $REPO/tools/ablate.py --cumulative
    python3 $REPO/tools/ablate_pmc_cum.py out/.../p_counter_collection.csv 10
"""
import collections, csv, sys
path, k = sys.argv[1], int(sys.argv[2])
import os, re
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ablate import SEGS
segs = [nm for _, nm in SEGS]
STEP = re.compile(r"k_env<\s*\d+,\s*[\w ]+,\s*true\s*,")     # k_env<N, dtype, DO_STEP = true, LAT>
rows = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if STEP.search(r["Kernel_Name"]):
        rows[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for cname, v in rows.items():
    v.sort()
    vals = [x[1] for x in v][100:]
    per = 5 + k
    print(cname)
    prev = 0.0
    for i, nm in enumerate(segs):
        chunk = vals[i * per + 5:(i + 1) * per]
        if not chunk:
            break
        m = sum(chunk) / len(chunk) / 4096
        print(f"  up to {nm:32s} {m:9.1f} per env   (+{m - prev:8.1f})")
        prev = m
