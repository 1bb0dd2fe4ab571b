#!/usr/bin/env python3
"""Diagnostic: where does one env-step kernel spend its cycles?  Builds/loads the -DSWARM_STAMPS library
(in-kernel clock64 stamps at phase boundaries, wave 0 of every workgroup) and prints the share of each phase.
Read SHARES, not absolute time: the stamps fence the scheduler (see cdna_hip_programming.md section 7).  The stamped wave's
role rotates with the workgroup index, so the shares are an average over the four roles; tools/ablate.py --cumulative is
the better instrument for throughput."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SWARM_LIB"] = os.path.join(ROOT, "marl_llm_amd", "lib", "libswarmenv_stamps.so")

import numpy as np
import torch

from marl_llm_amd.batched import SwarmBatch
from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
from marl_llm_amd.synth import synthetic_batch

PHASES = ["load cells+state", "forces+prior+integrate", "neighbour search", "cell scan", "occupied filter",
          "sensed list+reward", "obs stream"]


def main():
    n_a = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    shapes = synthetic_shape_set()
    ra = r_avoid_for(n_a, shapes)
    sy = synthetic_batch(E, n_a, shapes, seed=226)
    sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=ra)
    sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"]); sb.set_state(sy["p"], sy["dp"]); sb.observe()
    act = torch.zeros((E, n_a, 2), device=sb.device)
    for _ in range(100):
        act = sb.step(act)[3]
    fn = sb.lib.swarm_debug_stamps
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 5 + [ctypes.c_int]
    grid_max = E
    out = np.zeros((grid_max, 16), np.int64)
    obs, rew, done, pri = sb._obs[0], sb._rew[0], sb._done, sb._pri[0]
    g = fn(sb.handle, act.data_ptr(), 0, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), pri.data_ptr(),
           out.ctypes.data_as(ctypes.c_void_p), grid_max)
    assert g > 0, g
    t = out[:g].astype(np.float64)
    t = t[:, :8]
    d = np.diff(t, axis=1)
    tot = t[:, 7] - t[:, 0]
    print(f"{n_a} agents x {E} envs: {g} workgroups, mean cycles/workgroup {tot.mean():.0f} "
          f"(min {tot.min():.0f}, max {tot.max():.0f}); kernel span {(t[:, 7].max() - t[:, 0].min()):.0f} cycles")
    for k, name in enumerate(PHASES):
        print(f"  {name:26s} {d[:, k].mean():10.0f} cycles  {100 * d[:, k].mean() / tot.mean():5.1f} %")


if __name__ == "__main__":
    main()
