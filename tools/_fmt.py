import json, sys
d = json.loads(sys.stdin.read()); E = d["config"]["envs_per_gpu"]
print(sys.argv[1], E, round(d["value"] / 1e6, 1), "M/s kernel_us", round(d["roofline"]["kernel_us"], 1), "ns/env", round(d["roofline"]["kernel_us"] * 1000 / E, 2))
