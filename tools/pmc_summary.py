#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter CSVs (one counter group per pass) of `python bench.py` into the JSON that bench.py's
`roofline.traffic` reads: mean per launch of every counter over the timed k_env<..., true> launches, plus the HBM traffic
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes -- on gfx950 FETCH_SIZE counts a 128-byte read request as 64 bytes
(/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section), WRITE_SIZE is exact for streaming stores; both in KiB."""
import collections
import csv
import json
import sys

import re
STEP = re.compile(r"k_env<\s*\d+,\s*[\w ]+,\s*true\s*,")     # k_env<N, dtype, DO_STEP = true, LAT>: the step launches
out, last_n, workload = sys.argv[1], int(sys.argv[2]), sys.argv[3]
vals = collections.defaultdict(list)
for path in sys.argv[4:]:
    for r in csv.DictReader(open(path)):
        if STEP.search(r["Kernel_Name"]):
            vals[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
res = {}
for k, v in sorted(vals.items()):
    v.sort()
    tail = [x[1] for x in v][-last_n:]
    res[k] = {"launches": len(tail), "mean": sum(tail) / len(tail)}
if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
    res["_traffic_bytes_per_launch"] = (2 * res["FETCH_SIZE"]["mean"] + res["WRITE_SIZE"]["mean"]) * 1024
res["_workload"] = workload
res["_n_env"] = 4096
import hashlib, os
res["_source_sha256"] = hashlib.sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "marl_llm_amd", "csrc", "swarm_env.hip"), "rb").read()).hexdigest()[:16]
res["_note"] = ("rocprofv3 --pmc, one counter group per pass, last %d launches of k_env<64,float,true> in `python bench.py "
                "--no-cpu-baseline --steps %d --warmup 5`; FETCH_SIZE / WRITE_SIZE in KiB; traffic = (2*FETCH_SIZE + "
                "WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)" % (last_n, last_n))
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: (v["mean"] if isinstance(v, dict) else v) for k, v in res.items() if not k.startswith("_n")}, indent=1))
