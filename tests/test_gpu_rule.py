"""Rule-based expert controller (assembly.py:530-601) on the GPU vs the recorded reference
actions (tests/golden/g5_rule_*.npz) and vs the numpy restatement on synthetic batches.

Tolerance: 1e-12 absolute on the clipped action -- every term is fp64 in numpy's operation order, but np.cos is numpy's
vectorised routine and the device uses its own fp64 cos (a few ulp apart), and np.linalg.norm of a 1-D vector goes through
BLAS dot (FMA).  The step that consumes the action is the bit-exact path of test_gpu_parity.py."""
import numpy as np
import pytest

import os

from helpers import GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-12


def _batch(p, dp, grid, l_cell, r_avoid):
    import torch
    from marl_llm_amd.batched import SwarmBatch
    T, _, N = p.shape
    sb = SwarmBatch(n_env=T, n_agents=N, n_cells_max=grid.shape[1], r_avoid=float(r_avoid), device="cuda:0")
    sb.set_cells(np.repeat(grid[None], T, 0), np.full(T, grid.shape[1], np.int32), np.full(T, float(l_cell)))
    sb.set_state(p, dp)
    sb.observe()
    return sb


@pytest.mark.parametrize("n_a", [8, 32])
def test_rule_action_matches_recorded_reference(n_a):
    import torch
    z = load_golden(os.path.join(GOLDEN_DIR, f"g5_rule_n{n_a}.npz"))
    sb = _batch(z["p"], z["dp"], z["grid"], z["l_cell"], z["r_avoid"])
    u = sb.rule_action()                                               # [T, N, 2]
    got = u.cpu().numpy().transpose(0, 2, 1)                           # -> [T, 2, N]
    assert np.abs(got - z["u"]).max() <= TOL
    # the rule-mode step: feed the expert action back (f64) and land on the reference's next state
    sb.step(u)
    p1, dp1 = [x.cpu().numpy() for x in sb.get_state()]
    assert np.abs(p1 - z["p_next"]).max() <= 1e-13 and np.abs(dp1 - z["dp_next"]).max() <= 1e-12
    sb.close()


@pytest.mark.parametrize("n_a,n_env", [(30, 16), (64, 8)])
def test_rule_action_matches_restatement_on_synthetic_batches(n_a, n_env):
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
    from marl_llm_amd.synth import synthetic_batch
    from oracle.oracle_py import rule_action
    shapes = synthetic_shape_set()
    r_avoid = r_avoid_for(n_a, shapes)
    sy = synthetic_batch(n_env, n_a, shapes, seed=77)
    sb = SwarmBatch(n_env=n_env, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=r_avoid, device="cuda:0")
    sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"])
    sb.set_state(sy["p"], sy["dp"])
    sb.observe()
    for _ in range(30):                                                # assemble under the expert itself
        sb.step(sb.rule_action())
    p, dp = [x.cpu().numpy() for x in sb.get_state()]
    got = sb.rule_action().cpu().numpy().transpose(0, 2, 1)
    assert sb.indices(False, False)["in_flags"].float().mean().item() > 0.2      # the occupied-cell filter is exercised
    for e in range(n_env):
        g = np.ascontiguousarray(sy["cells"][e][:, : sy["n_g"][e]])
        want = rule_action(p[e], dp[e], g, float(sy["l_cell"][e]), r_avoid)
        assert np.abs(got[e] - want).max() <= TOL, e
    sb.close()


def test_env_rule_mode_returns_the_applied_action():
    """AssemblySwarmEnv(agent_strategy='rule', is_collected=True): step ignores the passed action and returns u."""
    from marl_llm_amd.env import AssemblySwarmWrapper, AssemblySwarmEnv, make_args
    from marl_llm_amd.shapes import synthetic_shape_set
    from oracle.oracle_py import rule_action
    np.random.seed(5)
    args = make_args(n_a=16, results_file=synthetic_shape_set(), agent_strategy="rule", is_collected=True)
    env = AssemblySwarmWrapper(AssemblySwarmEnv(), args)
    env.reset()
    b = env.env
    for _ in range(5):
        p, dp = [x.cpu().numpy()[0] for x in b._backend().get_state()]
        g = np.ascontiguousarray(b._cells[0][:, : b._n_g[0]])
        _, _, _, _, u = env.step(np.zeros((2, 16), np.float32))
        assert u.shape == (2, 16)
        assert np.abs(u - rule_action(p, dp, g, float(b._l_cell[0]), b.r_avoid)).max() <= TOL
