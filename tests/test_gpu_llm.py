"""agent_strategy == 'llm' (assembly.py:525-529): the agents are driven by robot_prior_policy, the Python twin of the prior
policy with repulsion gain 1.0 (assembly.py:892-940), evaluated on the device by the observation pass.  Checked against
steps the reference itself took in that mode (tests/golden/g10_llm_*.npz, recorded by make_golden.py r3 from the imported
reference env with is_collected=True so that step() returns the applied action u).

Tolerance 1e-12 absolute: every term is fp64 in the reference's operation order, but np.linalg.norm of a 2-vector goes
through BLAS dot (FMA), an ulp away from sqrt(x*x + y*y)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-12


@pytest.mark.parametrize("n_a", [8, 32])
def test_llm_action_and_step_match_recorded_reference(n_a):
    import torch
    from marl_llm_amd.batched import SwarmBatch
    z = load_golden(os.path.join(GOLDEN_DIR, f"g10_llm_n{n_a}.npz"))
    T, _, N = z["p"].shape
    grid = z["grid"]
    sb = SwarmBatch(n_env=T, n_agents=N, n_cells_max=grid.shape[1], r_avoid=float(z["r_avoid"]), device="cuda:0",
                    obs_dtype=torch.float64, llm_action=True)
    sb.set_cells(np.repeat(grid[None], T, 0), np.full(T, grid.shape[1], np.int32), np.full(T, float(z["l_cell"])))
    sb.set_state(z["p"], z["dp"])
    sb.observe()
    assert np.array_equal(sb.indices(False, False)["neighbor_index"].cpu().numpy(), z["nei_prev"])
    u = sb.llm_action().cpu().numpy().transpose(0, 2, 1)               # [T, 2, N]
    assert np.abs(u - z["u"]).max() <= TOL
    # the repulsion term (the only place the twin differs from the C++ prior) is active in the recorded steps
    close = 0
    for t in range(T):
        for i in range(N):
            js = z["nei_prev"][t, i]; js = js[js >= 0]
            d = np.linalg.norm(z["p"][t][:, js] - z["p"][t][:, [i]], axis=0)
            close += int(((d > 0) & (d < float(z["r_avoid"]))).sum())
    assert close > 0
    obs, rew, done, pri = sb.step(None)                                # the library applies its own llm action
    p1, dp1 = [x.cpu().numpy() for x in sb.get_state()]
    assert np.abs(p1 - z["p_next"]).max() <= TOL and np.abs(dp1 - z["dp_next"]).max() <= TOL
    got = obs.cpu().numpy().transpose(0, 2, 1)                         # [T, D, N]
    assert np.abs(got - z["obs"]).max() <= 1e-11
    assert np.array_equal(rew.cpu().numpy().astype(np.float64), z["rew"][:, 0, :])
    # a handle without the switch refuses a NULL action
    sb2 = SwarmBatch(n_env=1, n_agents=N, n_cells_max=grid.shape[1], r_avoid=float(z["r_avoid"]), device="cuda:0")
    with pytest.raises(Exception):
        sb2.step(None)
    sb.close(); sb2.close()


def test_llm_strategy_through_the_env_surface(shapes):
    """AssemblySwarmEnv(agent_strategy='llm', is_collected=True).step returns the applied action as the fifth element
    (assembly.py:663-664) -- numpy API and tensor API agree with the recorded reference step."""
    from marl_llm_amd.env import AssemblySwarmEnv, AssemblySwarmWrapper, make_args
    z = load_golden(os.path.join(GOLDEN_DIR, "g10_llm_n8.npz"))
    np.random.seed(1)
    env = AssemblySwarmWrapper(AssemblySwarmEnv(), make_args(n_a=8, results_file=shapes, agent_strategy="llm", is_collected=True))
    env.reset()
    base = env.env
    base.l_cell = float(z["l_cell"]); base.grid_center = z["grid"]; base.n_g = z["grid"].shape[1]   # eval_assembly.py:34-57 style
    assert abs(base.r_avoid - float(z["r_avoid"])) == 0.0
    t = 2
    base.set_state(z["p"][t], z["dp"][t])
    o, r, d, info, u = env.step(np.zeros((2, 8), np.float32))
    assert u.shape == (2, 8) and np.abs(u - z["u"][t]).max() <= TOL
    assert np.abs(base.p - z["p_next"][t]).max() <= TOL
    assert o.shape == (192, 8) and o.dtype == np.float64 and np.abs(o - z["obs"][t]).max() <= 1e-11
    assert np.array_equal(r, z["rew"][t])
    env.close()
