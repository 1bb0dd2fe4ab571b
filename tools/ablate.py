#!/usr/bin/env python3
"""Diagnostic: time the step kernel with one phase run EXTRA times (debug_flags bits 8..11 = phase, 12..15 = extra
repeats; the phases are idempotent, so outputs do not change).  The extra time / instructions per repeat are that
phase's cost.  Under `rocprofv3 --pmc SQ_INSTS_VALU` + tools/ablate_pmc.py this gives exact per-phase VALU counts."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from marl_llm_amd.batched import SwarmBatch
from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
from marl_llm_amd.synth import synthetic_batch

NAMES = ["forces+prior+integrate", "pair masks", "cell scan", "occupied filter", "rank-select bits", "reward sums",
         "obs head pairs", "obs sensed pairs", "cell staging", "ordered insertion", "emit walk", "nearest merge"]


# (exit code, label) in program order of the lattice (row-space) kernel; the early exits' TIMES below ~10 us are bounded by the
# host's launch rate, their instruction counts are exact
SEGS = [(0, "loads + integration + first barrier"), (1, "(state re-read)"), (2, "pair masks"),
        (3, "contact spring + ordered insertion (B)"), (4, "cell walk"), (5, "nearest merge"), (6, "kept rows + counts"),
        (7, "rank by list length"), (9, "list emission + reward sums"), (10, "reward verdict (in-wave)"),
        (11, "prior (B) | obs heads (A, C)"), (None, "obs rows of the waves' own agents (full kernel)")]


def run(skip, n_a, E, sy, ra, state, steps=int(os.environ.get('ABLATE_STEPS', '60'))):
    sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=ra, debug_flags=skip)
    sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"])
    sb.set_state(state[0], state[1]); sb.observe()
    act = torch.zeros((E, n_a, 2), device=sb.device)
    for _ in range(5):
        sb.step(act)
    sb.set_state(state[0], state[1]); sb.observe()
    sb.timer_start()
    for _ in range(steps):
        sb.step(act)               # zero action: the swarm stays where it is -> same work every step
    ms = sb.timer_stop() / steps
    sb.close()
    return ms


def main():
    dbg = sys.argv[sys.argv.index("--dbg") + 1] if "--dbg" in sys.argv else None
    pos = [a for a in sys.argv[1:] if not a.startswith('--') and a != dbg]
    n_a = int(pos[0]) if len(pos) > 0 else 64
    E = int(pos[1]) if len(pos) > 1 else 4096
    shapes = synthetic_shape_set()
    ra = r_avoid_for(n_a, shapes)
    sy = synthetic_batch(E, n_a, shapes, seed=226)
    sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=ra)
    sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"]); sb.set_state(sy["p"], sy["dp"]); sb.observe()
    act = torch.zeros((E, n_a, 2), device=sb.device)
    for _ in range(100):
        act = sb.step(act)[3]
    state = [x.cpu().numpy() for x in sb.get_state()]
    sb.close()
    if "--dbg" in sys.argv:      # full kernel with one experiment switch (library built with -DSWARM_EXPERIMENT)
        for tok in ["0"] + dbg.split(","):                    # "phase" or "phase:extra"
            ph, ex = (int(v) for v in (tok.split(":") + ["0"])[:2])
            print(f"  debug phase {ph:2d} extra {ex:2d}: {run((ph << 8) | (ex << 12), n_a, E, sy, ra, state) * 1e3:8.1f} us")
        return
    if "--cumulative" in sys.argv:
        # leave the kernel after segment k (debug phase 15): cumulative time / counters up to each point
        # (exit code, label): the lattice (row-space) kernel has no separate rank-select / reward-sum segments
        prev = 0.0
        for code, nm in SEGS:
            ms = run((15 << 8) | (code << 12), n_a, E, sy, ra, state) if code is not None else run(0, n_a, E, sy, ra, state)
            print(f"  exit after {nm:32s} {ms * 1e3:8.1f} us   (+{(ms - prev) * 1e3:6.1f})")
            prev = ms
        return
    extra = 2
    base = run(0, n_a, E, sy, ra, state)
    print(f"{n_a} agents x {E} envs, assembled, zero action: full kernel {base * 1e3:.1f} us")
    tot = 0.0
    for k, nm in enumerate(NAMES):
        ms = run(((k + 1) << 8) | (extra << 12), n_a, E, sy, ra, state)
        d = (ms - base) / extra
        tot += d
        print(f"  {nm:24s} +{extra} repeats: {ms * 1e3:8.1f} us   (phase ~ {d * 1e3:6.1f} us, {100 * d / base:5.1f} %)")
    print(f"  sum of phases {tot * 1e3:.1f} us; remainder (loads, staging, barriers, merges) {(base - tot) * 1e3:.1f} us")


if __name__ == "__main__":
    main()
