#!/usr/bin/env python3
"""Diagnostic: where does one env-step kernel spend its cycles, per wave ROLE?  Loads the -DSWARM_STAMPS library
(marl_llm_amd.build.build_lib(stamps=True): in-kernel clock64 stamps at every barrier and phase boundary, one record per
wave) and prints, for each role (split A = forces / reward combine, B = prior / insertion, C and D = walk only), the mean
cycles between consecutive stamp points.  Read SHARES, not absolute time: the stamps fence the scheduler
(cdna_hip_programming.md section 7).  Usage: python3 tools/phase_profile.py [agents] [envs]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SWARM_LIB"] = os.environ.get("SWARM_STAMPS_LIB", os.path.join(ROOT, "marl_llm_amd", "lib", "libswarmenv_stamps.so"))

import numpy as np
import torch

from marl_llm_amd.batched import SwarmBatch
from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
from marl_llm_amd.synth import synthetic_batch

# stamp ids in program order and what ENDS at each of them
# (lattice / row-space kernel: the synthetic and the reference's shapes; the generic scan keeps the round-2 stamp points)
ORDER = [(0, "start"), (11, "A: state / action / mask loads landed"), (10, "A: forces"), (12, "A: integrate + publish"),
         (1, "first barrier (others: lattice rows -> LDS, waiting)"), (2, "forces | prior, integrate, 2 barriers"),
         (14, "pair-mask loop"), (15, "pair masks: LDS exchange + barrier"), (3, "ordered insertion (B only)"),
         (16, "lattice walk (not B)"), (17, "barrier after walk"), (4, "nearest merge (LDS)"),
         (5, "kept rows + counts + list fill + barrier"), (18, "rank by list length + barrier"),
         (19, "list emission + reward sums (quads)"), (6, "reward verdict + store (in-wave)"),
         (9, "prior (B) | obs heads (A, C)"), (7, "obs rows of the wave's own 16 agents")]


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
    npad = max(8, 1 << (n_a - 1).bit_length())
    wpb = max(64, npad) * 4 // 64
    epb = 64 // npad if npad < 64 else 1
    grid = (E + epb - 1) // epb
    out = np.zeros((grid, wpb, 24), np.int64)
    obs, rew, done, pri = sb._obs[0], sb._rew[0], sb._done, sb._pri[0]
    g = fn(sb.handle, act.data_ptr(), 0, obs.data_ptr(), rew.data_ptr(), done.data_ptr(), pri.data_ptr(),
           out.ctypes.data_as(ctypes.c_void_p), grid)
    assert g == grid, g
    t = out.reshape(-1, 24).astype(np.float64)
    role = out.reshape(-1, 24)[:, 23]
    span = t[:, 7].max() - t[:, 0].min()
    life = t[:, 7] - t[:, 0]
    print(f"{n_a} agents x {E} envs: {grid} workgroups x {wpb} waves; kernel span {span:.0f} cycles; "
          f"mean wave life {life.mean():.0f} cycles (min {life.min():.0f}, max {life.max():.0f})")
    names = {0: "A (forces, reward combine)", 1: "B (prior, insertion)", 2: "C (walk)", 3: "D (walk)"}
    ids = [k for k, _ in ORDER]
    print(f"{'segment':44s}" + "".join(f"{names[r][:1]:>10s}" for r in range(4)) + "      all   share")
    tot_all = life.mean()
    for q in range(1, len(ORDER)):
        k, nm = ORDER[q]
        cols = []
        for r in range(4):
            m = role == r
            # a stamp a role never reaches (value 0) inherits the previous one: carry forward
            cur = t[m][:, ids[: q + 1]].copy()
            for c in range(1, cur.shape[1]):
                z = cur[:, c] == 0
                cur[z, c] = cur[z, c - 1]
            cols.append((cur[:, q] - cur[:, q - 1]).mean())
        allm = float(np.mean(cols))
        print(f"{nm:44s}" + "".join(f"{c:10.0f}" for c in cols) + f"{allm:9.0f}  {100 * allm / tot_all:5.1f} %")


if __name__ == "__main__":
    main()
