#!/usr/bin/env python3
"""Consistency stress (GPU): free-running batches with noisy prior actions through the four code-path combinations --
fast / every exact fallback forced (debug_flags bit 0) x lattice walk / generic scan (bit 1) -- must produce bit-identical
observations, rewards, priors, state and index scratch.  Complements tests/test_gpu_parity.py (which checks against the
oracle at sizes the oracle finishes in seconds) at millions of agent-steps."""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from marl_llm_amd.batched import SwarmBatch
from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
from marl_llm_amd.synth import synthetic_batch
shapes = synthetic_shape_set()
def run(n_a, E, flags, steps, seed, frac, g_max=80):
    ra = r_avoid_for(n_a, shapes)
    sy = synthetic_batch(E, n_a, shapes, seed=seed, assembled_fraction=frac)
    sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=ra, debug_flags=flags, g_max=g_max, device="cuda:0")
    sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"]); sb.set_state(sy["p"], sy["dp"]); sb.observe()
    gen = torch.Generator(device="cuda").manual_seed(seed)
    act = torch.zeros((E, n_a, 2), device="cuda")
    acc = []
    for t in range(steps):
        obs, rew, done, pri = sb.step(act)
        act = pri if t % 3 else (pri + 0.3 * torch.randn(pri.shape, device="cuda", generator=gen)).clamp(-1, 1)
        acc.append((obs.clone(), rew.clone(), pri.clone()))
    p, dp = sb.get_state(); idx = sb.indices()
    out = (torch.stack([a[0] for a in acc[-3:]]), torch.stack([a[1] for a in acc]), torch.stack([a[2] for a in acc[-3:]]), p, dp,
           idx["sensed_index"], idx["occupied_index"], idx["neighbor_index"], idx["in_flags"])
    sb.close()
    return out
ok = True
CASES = [(64, 2048, 40, 0.6, 80), (64, 1024, 25, 0.0, 80), (30, 2048, 30, 0.7, 80), (64, 512, 25, 0.8, 24), (128, 256, 12, 0.6, 80), (8, 2048, 30, 0.5, 80), (64, 256, 20, 0.8, 25)]
SEED = 7
if "--extended" in sys.argv:      # the other workgroup geometries (N = 16, 100, 200, 256; the half-occupied one: 32 x 256) and another seed
    CASES = [(256, 96, 10, 0.6, 80), (200, 64, 10, 0.7, 80), (100, 256, 15, 0.6, 80), (16, 2048, 30, 0.6, 80), (32, 256, 30, 0.7, 80), (64, 4096, 30, 0.7, 80), (128, 512, 12, 0.3, 80)]
    SEED = 11
for (n_a, E, steps, frac, g_max) in CASES:
    ref = run(n_a, E, 0, steps, SEED, frac, g_max)
    for flags in (1, 2, 3):
        got = run(n_a, E, flags, steps, SEED, frac, g_max)
        same = all(torch.equal(a, b) for a, b in zip(ref, got))
        ok &= same
        print(f"N={n_a} E={E} steps={steps} frac={frac} G={g_max} flags={flags}: {'identical' if same else 'MISMATCH'}  reward sum {ref[1].sum().item():.0f}", flush=True)
print("ALL IDENTICAL" if ok else "FAILURES")
