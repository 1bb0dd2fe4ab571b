"""Caller-shaped drop-in test: the exact call sequence of the reference's train / eval scripts against our env.

train_assembly.py:49-111,132,144 -- make -> .unwrapped -> AssemblySwarmWrapper(base_env, args) -> observation_space /
action_space / num_agents -> reset -> [render, step, np.mean(rewards)] x k -> env.alpha read -> env.env.alpha = 0.1.
eval_assembly.py:137-162 -- np.shape(env.p) -> [render, env.p / env.dp reads, process_shape() attribute writes
(:34-57) at the switch times, the three wrapper metrics BEFORE the next step, step].
Every step is checked against the oracle (bit for bit), the post-switch metrics against oracle_py.wrapper_metrics on the
NEW cells (that function is pinned exactly against the reference's own wrapper by tests/test_oracle_golden.py)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _make():
    """gym.make('AssemblySwarm-v0') when a gym package is importable, else this package's stand-in of it."""
    from marl_llm_amd import env as E
    try:
        import gym
        E.register(gym)
        return gym.make("AssemblySwarm-v0")
    except ImportError:
        return E.make("AssemblySwarm-v0")


def process_shape(shape_index, env, l_cells_input, grid_center_origins_input, binary_images_input,
                  shape_bound_points_origins_input):
    """The attribute writes of eval_assembly.py:34-57 in their order (rotation 0, offset 0 as there)."""
    env.env.l_cell = l_cells_input[shape_index]
    env.env.grid_center_origin = grid_center_origins_input[shape_index].T
    env.env.target_shape = binary_images_input[shape_index]
    env.env.shape_bound_points_origin = shape_bound_points_origins_input[shape_index]
    rotate_matrix = np.array([[np.cos(0), np.sin(0)], [-np.sin(0), np.cos(0)]])
    env.env.grid_center_origin = np.dot(rotate_matrix, env.env.grid_center_origin)
    env.env.n_g = env.env.grid_center_origin.shape[1]
    rand_target_offset = np.zeros((2, 1))
    env.env.grid_center = env.env.grid_center_origin.copy() + rand_target_offset
    env.env.shape_bound_points = np.hstack((env.env.shape_bound_points_origin[:2] + rand_target_offset[0, 0],
                                            env.env.shape_bound_points_origin[2:] + rand_target_offset[1, 0]))


@pytest.mark.parametrize("n_a", [30, 64])
def test_train_and_eval_call_sequence(n_a, shapes, oracle):
    from marl_llm_amd.env import AssemblySwarmWrapper, make_args
    from oracle.oracle_py import wrapper_metrics
    args = make_args(n_a=n_a, results_file=shapes)            # cfg/assembly_cfg.py:153-168 defaults
    np.random.seed(226)                                       # train_assembly.py:37 (cfg.seed)
    base_env = _make().unwrapped                              # :49
    env = AssemblySwarmWrapper(base_env, args)                # :50
    start_stop_num = [slice(0, env.num_agents)]               # :51
    assert env.num_agents == n_a and env.agent_types == ["agent"] and len(env.agents) == n_a
    assert env.observation_space.shape == (192, n_a) and env.action_space.shape == (2, n_a)   # :66-69, maddpg.py:264-266

    obs = env.reset()                                         # :81
    assert obs.shape == (192, n_a) and obs.dtype == np.float64
    start_stop_num = [slice(0, env.n_a)]                      # :82
    ra = env.r_avoid
    grid = env.grid_center.copy(); l_cell = env.l_cell
    p, dp = env.p, env.dp
    o0 = oracle.get_observation(p, dp, grid, l_cell, ra)
    assert np.array_equal(obs, o0["obs"])
    nei = o0["neighbor_index"]
    rng = np.random.RandomState(1)
    ep_mean = 0.0
    for et_index in range(6):                                 # :91
        assert env.render() is None                           # :94-95
        torch_obs = torch.Tensor(obs)                         # :98
        assert torch_obs[:, start_stop_num[0]].t().shape == (n_a, 192)         # maddpg.py:84
        agent_actions = rng.uniform(-1, 1, (2, n_a)).astype(np.float32)       # stands in for maddpg.step (:99-100)
        next_obs, rewards, dones, _, agent_actions_prior = env.step(agent_actions)   # :102
        s = oracle.step(p, dp, agent_actions.astype(np.float64), grid, nei, l_cell, ra)
        assert np.array_equal(next_obs, s["obs"]) and np.array_equal(rewards, s["reward"])
        assert dones.shape == (1, n_a) and dones.dtype == bool and not dones.any()
        assert np.array_equal(agent_actions_prior, s["a_prior"])
        p, dp, nei = s["p"], s["dp"], s["neighbor_index"]
        obs = next_obs
        ep_mean += np.mean(rewards)                           # :110
    assert env.alpha == 1                                     # :132
    env.env.alpha = 0.1                                       # :144
    assert env.alpha == 0.1

    # ---- eval_assembly.py:137-162
    M_p, N_p = np.shape(env.p); M_v, N_v = np.shape(env.dp)   # :137-138
    assert (M_p, N_p) == (2, n_a) and (M_v, N_v) == (2, n_a)
    l_cells, origins = shapes["l_cell"], shapes["grid_coords"]
    for et_index, switch_to in enumerate([4, None, 5, None]):
        assert env.render() is None                           # :147
        assert np.array_equal(env.p, p) and np.array_equal(env.dp, dp)        # :150-151
        if switch_to is not None:                             # :154-157
            process_shape(switch_to, env, l_cells, origins, shapes["binary_image"], shapes["shape_bound_points"])
            grid = np.ascontiguousarray(origins[switch_to].T); l_cell = float(l_cells[switch_to])
            nei = oracle.get_observation(p, dp, grid, l_cell, ra)["neighbor_index"]
        # the three metrics right after the switch, before any step (:160-162): they must see the NEW cells
        m = wrapper_metrics(p, grid, ra)
        got = np.array([env.coverage_rate(), env.distribution_uniformity(), env.voronoi_based_uniformity()])
        assert np.array_equal(got, m, equal_nan=True), (et_index, got, m)
        ind = env.env.indices()
        o = oracle.get_observation(p, dp, grid, l_cell, ra)
        for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
            assert np.array_equal(ind[k][0], o[k]), k
        a = rng.uniform(-1, 1, (2, n_a)).astype(np.float32)
        next_obs, rewards, dones, _, pri = env.step(a)        # eval :165-170
        s = oracle.step(p, dp, a.astype(np.float64), grid, nei, l_cell, ra)
        assert np.array_equal(next_obs, s["obs"]) and np.array_equal(rewards, s["reward"]) and np.array_equal(pri, s["a_prior"])
        p, dp, nei = s["p"], s["dp"], s["neighbor_index"]
    env.close()


def test_metrics_see_a_shape_switch_without_a_step(shapes, oracle):
    """ADVICE r1 (stale cells): assign l_cell / n_g / grid_center as eval does and read a metric before any step."""
    from marl_llm_amd.env import AssemblySwarmEnv, AssemblySwarmWrapper, make_args
    from oracle.oracle_py import wrapper_metrics
    np.random.seed(3)
    env = AssemblySwarmWrapper(AssemblySwarmEnv(), make_args(n_a=16, results_file=shapes))
    env.reset()
    p = env.p
    old = env.coverage_rate()
    g_new = np.ascontiguousarray(shapes["grid_coords"][2].T) + np.array([[p[0].mean()], [p[1].mean()]])   # under the swarm
    env.env.l_cell = shapes["l_cell"][2]; env.env.n_g = g_new.shape[1]; env.env.grid_center = g_new
    m = wrapper_metrics(p, g_new, env.r_avoid)
    assert env.coverage_rate() == m[0] and env.voronoi_based_uniformity() == m[2]
    assert m[0] != old or m[0] == 0.0
    env.close()


def test_device_reset_mode_exposes_l_cell_and_shape_index(shapes):
    """rng='device' must leave l_cell / shape_index / shape_frequency usable (a later shape switch that
    assigns only grid_center re-uploads with the per-env l_cell)."""
    from marl_llm_amd.env import AssemblySwarmEnv, AssemblySwarmWrapper, make_args
    env = AssemblySwarmWrapper(AssemblySwarmEnv(n_envs=8, rng="device", seed=5), make_args(n_a=16, results_file=shapes))
    env.reset()
    e = env.env
    assert e.shape_index.shape == (8,) and (e.shape_index >= 0).all() and (e.shape_index < 7).all()
    assert np.array_equal(e._l_cell, np.asarray(shapes["l_cell"])[e.shape_index]) and e.l_cell > 0
    assert np.array_equal(e._n_g, np.asarray([g.shape[0] for g in shapes["grid_coords"]])[e.shape_index])
    assert e.shape_frequency.sum() == 8
    e.grid_center = np.ascontiguousarray(shapes["grid_coords"][1].T)     # only the cells: l_cell stays per env
    obs, rew, done, info, pri = env.step(np.zeros((2, 8 * 16), np.float32))
    assert np.isfinite(obs).all()
    env.close()
