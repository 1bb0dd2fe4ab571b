"""The reference-shaped host boundary (swarm_step_host / swarm_observe_host): the numpy API of
AssemblySwarmEnv.step / reset returns the reference's shapes and dtypes (assembly.py:487-666: obs (D, n_a) f64, reward
(1, n_a) f64, done (1, n_a) bool, a_prior (2, n_a) f64) and must carry exactly the values of the device-tensor API --
the widening / transposition happens on the device, the data lands in pinned memory with one copy."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _pair(shapes, n_envs, n_a, obs_dtype, host_copy, **over):
    from marl_llm_amd.env import AssemblySwarmEnv, make_args
    envs = []
    for hc in (host_copy, "auto"):
        e = AssemblySwarmEnv(n_envs=n_envs, obs_dtype=obs_dtype, rng="counter", seed=5, host_copy=hc)
        e.__reinit__(make_args(n_a=n_a, results_file=shapes, **over))
        envs.append(e)
    return envs


@pytest.mark.parametrize("n_envs,n_a,obs_dtype,is_boundary", [(1, 30, "float64", True), (5, 16, "float64", True),
                                                             (3, 64, "float32", True), (4, 8, "float64", False)])
def test_numpy_api_equals_tensor_api(shapes, n_envs, n_a, obs_dtype, is_boundary):
    ea, eb = _pair(shapes, n_envs, n_a, obs_dtype, False, is_boundary=is_boundary)
    E, N = n_envs, n_a
    o_np = ea.reset()
    o_t = eb.reset_tensor()
    D = ea.obs_dim_agent
    assert o_np.shape == (D, E * N) and o_np.dtype == np.float64
    assert np.array_equal(o_np, o_t.reshape(E * N, D).cpu().numpy().astype(np.float64).T)
    rng = np.random.default_rng(0)
    keep = []
    for t in range(6):
        a = rng.uniform(-1, 1, (2, E * N)).astype(np.float32 if t % 2 else np.float64)
        obs, rew, done, info, pri = ea.step(a)
        at = torch.as_tensor(np.ascontiguousarray(a.T.reshape(E, N, 2)), device=eb._backend().device)
        obs_t, rew_t, done_t, pri_t = eb.step_tensor(at)
        assert obs.shape == (D, E * N) and obs.dtype == np.float64
        assert rew.shape == (1, E * N) and rew.dtype == np.float64
        assert done.shape == (1, E * N) and done.dtype == np.bool_ and not done.any()
        assert pri.shape == (2, E * N) and pri.dtype == np.float64 and info.shape == (3, 1)
        assert np.array_equal(obs, obs_t.reshape(E * N, D).cpu().numpy().astype(np.float64).T)
        assert np.array_equal(rew[0], rew_t.reshape(-1).cpu().numpy().astype(np.float64))
        assert np.array_equal(pri, pri_t.reshape(E * N, 2).cpu().numpy().astype(np.float64).T)
        assert np.array_equal(ea.p, eb.p) and np.array_equal(ea.dp, eb.dp)
        keep.append(obs)
    # host_copy=False hands out views of two alternating pinned slots: consecutive results never share memory (the trainer
    # holds obs and next_obs at once, train_assembly.py:97-111), every second one does
    assert not np.shares_memory(keep[-1], keep[-2]) and np.shares_memory(keep[-1], keep[-3])
    ea.close(); eb.close()


def test_small_outputs_are_copies_by_default(shapes):
    from marl_llm_amd.env import AssemblySwarmEnv, make_args
    e = AssemblySwarmEnv(n_envs=1, rng="counter", seed=2)
    e.__reinit__(make_args(n_a=30, results_file=shapes))
    o0 = e.reset()
    first = o0.copy()
    outs = [e.step(np.zeros((2, 30)))[0] for _ in range(3)]
    assert np.array_equal(o0, first)                                   # untouched by later steps
    assert not any(np.shares_memory(outs[0], o) for o in outs[1:])
    e.close()
