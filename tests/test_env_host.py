"""Host logic of the gym-surface env that needs no GPU: reset() draw order against the golden vector recorded
from the reference under seed 226 (tests/golden/g4_reset_seed226_n8.npz), spaces, argument handling."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, load_golden


def _env(shapes, n_a=8, n_envs=1, **kw):
    from marl_llm_amd.env import AssemblySwarmEnv, AssemblySwarmWrapper, make_args
    env = AssemblySwarmEnv(n_envs=n_envs)
    return AssemblySwarmWrapper(env, make_args(n_a=n_a, results_file=shapes, **kw))


def test_reset_draw_order_matches_reference(shapes):
    z = load_golden(os.path.join(GOLDEN_DIR, "g4_reset_seed226_n8.npz"))
    np.random.seed(226)
    env = _env(shapes)                     # __reinit__ consumes n_a^2 draws like assembly.py:133
    s = env.env._sample_reset()
    assert np.array_equal(s["p"][0], z["p"]) and np.array_equal(s["dp"][0], z["dp"])
    ng = int(s["n_g"][0])
    assert ng == z["grid"].shape[1]
    np.testing.assert_allclose(s["cells"][0][:, :ng], z["grid"], rtol=0, atol=1e-15)   # numpy BLAS 2x2 dot
    assert float(s["l_cell"][0]) == float(z["l_cell"])
    assert env.r_avoid == float(z["r_avoid"])
    assert np.array_equal(env.env.shape_frequency, z["shape_frequency"])


def test_surface_and_spaces(shapes):
    env = _env(shapes, n_a=30)
    assert env.num_agents == 30 and env.agent_types == ["agent"] and len(env.agents) == 30
    assert env.observation_space.shape == (192, 30) and env.action_space.shape == (2, 30)
    assert env.n_a == 30 and env.alpha == 1
    env.env.alpha = 0.1                     # train_assembly.py:144 writes through env.env
    assert env.alpha == 0.1
    env2 = _env(shapes, n_a=30, is_con_self_state=False)
    assert env2.observation_space.shape == (188, 30)
    flat = _env(shapes, n_a=16, n_envs=4)   # 4 envs presented as one env with 64 agents
    assert flat.n_a == 64 and flat.observation_space.shape == (192, 64)
    assert flat.r_avoid == _env(shapes, n_a=16).r_avoid       # r_avoid from the per-env agent count


def test_unsupported_modes_fail_loudly(shapes):
    with pytest.raises(ValueError):
        _env(shapes, dynamics_mode="Polar")
    with pytest.raises(ValueError):
        _env(shapes, agent_strategy="greedy")      # the reference's step prints 'Wrong in Step function' (assembly.py:602-603)
    assert _env(shapes, agent_strategy="llm").agent_strategy == "llm"      # all four of the reference's strategies configure
