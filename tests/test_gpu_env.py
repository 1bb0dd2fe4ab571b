"""The gym-surface env on the GPU: reference shapes / dtypes, the golden trajectories through reset-less state
injection, and the flattened multi-env presentation."""
import os

import numpy as np
import pytest

from helpers import fig_shapes, golden_files, load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _env(results, n_a, n_envs=1, **kw):
    from marl_llm_amd.env import AssemblySwarmEnv, AssemblySwarmWrapper, make_args
    return AssemblySwarmWrapper(AssemblySwarmEnv(n_envs=n_envs), make_args(n_a=n_a, results_file=results, **kw))


@pytest.mark.parametrize("path", [p for p in golden_files() if "n256" not in p], ids=os.path.basename)
def test_golden_trajectory_through_env_api(path, shapes):
    """Free-running: inject the first recorded state, then step with the recorded actions; every output of
    every step equals the reference's (float64 obs mode), i.e. the whole trajectory is reproduced."""
    z = load_golden(path)
    T, _, n_a = z["p"].shape
    results = fig_shapes() if os.path.basename(path).startswith("g6_") else shapes      # g6: the reference's own fig/*.png shapes
    env = _env(results, n_a, is_boundary=bool(z["is_boundary"]), is_con_self_state=bool(z["with_self"]))
    env.reset()
    e = env.env
    e.r_avoid = float(z["r_avoid"])
    e.grid_center = z["grid"]; e.l_cell = float(z["l_cell"])
    assert e.r_avoid == env.r_avoid
    e.set_state(z["p"][0], z["dp"][0])
    for t in range(T):
        obs, rew, done, info, pri = env.step(z["a"][t])
        assert obs.shape == z["obs"][t].shape and obs.dtype == np.float64
        assert np.array_equal(obs, z["obs"][t]), t
        assert np.array_equal(rew, z["rew"][t]) and rew.shape == (1, n_a)
        assert np.array_equal(done, z["done"][t]) and done.dtype == bool
        assert np.array_equal(pri, z["a_prior"][t])
        assert info.shape == (3, 1)
        assert np.array_equal(e.p, z["p_next"][t]) and np.array_equal(e.dp, z["dp_next"][t])
    env.close()


def test_flattened_multi_env_equals_separate_envs(shapes):
    """E envs presented as one env with n_a = E*N (what the unchanged trainer sees) == E single envs."""
    E, N = 3, 16
    np.random.seed(5)
    flat = _env(shapes, N, n_envs=E)
    obs0 = flat.reset()
    assert obs0.shape == (192, E * N)
    p0, dp0 = flat.env.p, flat.env.dp
    cells, n_g, l_cell = flat.env._cells.copy(), flat.env._n_g.copy(), flat.env._l_cell.copy()
    a = np.random.uniform(-1, 1, (2, E * N)).astype(np.float32)
    obs1, rew1, done1, _, pri1 = flat.step(a)
    for k in range(E):
        one = _env(shapes, N)
        one.reset()
        e = one.env
        e.grid_center = cells[k][:, : n_g[k]]; e.l_cell = float(l_cell[k])
        sl = slice(k * N, (k + 1) * N)
        o0 = e.set_state(p0[:, sl], dp0[:, sl])
        assert np.array_equal(o0, obs0[:, sl])
        o, r, d, _, pr = one.step(a[:, sl])
        assert np.array_equal(o, obs1[:, sl]) and np.array_equal(r, rew1[:, sl]) and np.array_equal(pr, pri1[:, sl])
        one.close()
    flat.close()


def test_tensor_api_shapes(shapes):
    E, N = 5, 32
    env = _env(shapes, N, n_envs=E).env
    env._obs_dtype = "float32"; env._batch = None
    obs = env.reset_tensor()
    assert obs.shape == (E, N, 192) and obs.dtype == torch.float32 and obs.is_cuda
    act = torch.zeros((E, N, 2), device=obs.device)
    obs, rew, done, pri = env.step_tensor(act)
    assert rew.shape == (E, N) and done.shape == (E, N) and done.dtype == torch.uint8 and pri.shape == (E, N, 2)
    env.close()


def test_device_reset_mode_of_the_env(shapes):
    """rng='device': reset() without host-side sampling; same surface, reproducible per (seed, episode)."""
    from marl_llm_amd.env import AssemblySwarmEnv, AssemblySwarmWrapper, make_args
    def mk():
        return AssemblySwarmWrapper(AssemblySwarmEnv(n_envs=4, rng="device", seed=77), make_args(n_a=16, results_file=shapes))
    a, b = mk(), mk()
    o1, o2 = a.reset(), b.reset()
    assert o1.shape == (192, 64) and np.array_equal(o1, o2)
    o1b = a.reset()                       # next episode: new draw
    assert not np.array_equal(o1, o1b)
    act = np.zeros((2, 64), np.float32)
    obs, rew, done, info, pri = a.step(act)
    assert obs.shape == (192, 64) and rew.shape == (1, 64) and pri.shape == (2, 64)
    assert 0.0 <= a.coverage_rate() <= 1.0
    a.close(); b.close()
