"""SURVEY.md section 8f rows built beyond the step itself: batched device-side reset and the device-resident rollout.

Reset parity: the reference's reset() draws from numpy's global RNG, so a batched device reset cannot be seed-for-seed
identical to it; its ALGORITHM (assembly.py:156-219) is restated below in numpy on the same counter-based generator and
compared with the device: agent states bit-exact (pure fp64 multiply-add), cells to 4e-16 relative (device sin/cos vs
libm differ by <= 1 ulp); then the first observation equals the oracle's on the state the device produced."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GOLD = 0x9E3779B97F4A7C15
M64 = (1 << 64) - 1


def mix64(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def u01(key, k):
    return float(mix64(key + GOLD * (k + 1)) >> 11) * (1.0 / 9007199254740992.0)


def reset_reference(seed, episode, env_id, shapes, n_a, W=2.4, H=2.4):
    key = mix64(mix64(seed + GOLD * (episode + 1)) ^ env_id)
    S = len(shapes["l_cell"])
    s = min(int(u01(key, 0) * S), S - 1)
    ang = np.pi * (2.0 * u01(key, 1) - 1.0)
    cs, sn = np.cos(ang), np.sin(ang)
    offx = (-W + 1) + u01(key, 4) * (2 * W - 2); offy = (-H + 1) + u01(key, 5) * (2 * H - 2)
    g0 = np.asarray(shapes["grid_coords"][s], np.float64).T
    gx = cs * g0[0] + sn * g0[1] + offx; gy = -sn * g0[0] + cs * g0[1] + offy
    spread = (2.0 * u01(key, 6) - 1.0) > 0
    cx = (-W + 1) + u01(key, 7) * (2 * W - 2); cy = (-H + 1) + u01(key, 8) * (2 * H - 2)
    p = np.zeros((2, n_a)); dp = np.zeros((2, n_a))
    for i in range(n_a):
        ux, uy = u01(key, 16 + i), u01(key, 16 + n_a + i)
        if spread:
            p[0, i] = -W + ux * (2 * W); p[1, i] = -H + uy * (2 * H)
        else:
            p[0, i] = (2.0 * ux - 1.0) + cx; p[1, i] = (2.0 * uy - 1.0) + cy
        dp[0, i] = -0.5 + u01(key, 16 + 2 * n_a + i); dp[1, i] = -0.5 + u01(key, 16 + 3 * n_a + i)
    return s, np.stack([gx, gy]), p, dp


@pytest.mark.parametrize("n_a,n_env,env_offset", [(64, 48, 0), (30, 17, 1000), (256, 3, 7)])
def test_device_reset_matches_its_restatement_and_oracle(oracle, shapes, n_a, n_env, env_offset):
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.shapes import r_avoid_for
    ra = r_avoid_for(n_a, shapes)
    ng_max = max(np.asarray(g).shape[0] for g in shapes["grid_coords"])
    sb = SwarmBatch(n_env=n_env, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra, obs_dtype=torch.float64)
    sb.set_shapes(shapes)
    obs = sb.reset(seed=226, episode=3, env_offset=env_offset).cpu().numpy()
    assert sb.lattice_envs() == n_env
    p, dp = [x.cpu().numpy() for x in sb.get_state()]
    cells, n_g = sb.get_cells()
    idx = sb.indices()
    shape_hist = np.zeros(len(shapes["l_cell"]), int)
    for e in range(n_env):
        s, g, pr, dpr = reset_reference(226, 3, env_offset + e, shapes, n_a)
        shape_hist[s] += 1
        assert n_g[e] == g.shape[1]
        assert np.array_equal(p[e], pr) and np.array_equal(dp[e], dpr)
        np.testing.assert_allclose(cells[e][:, : n_g[e]], g, rtol=0, atol=2e-15)
        # first observation: the oracle on exactly the state / cells the device holds
        ge = np.ascontiguousarray(cells[e][:, : n_g[e]])
        o = oracle.get_observation(p[e], dp[e], ge, float(shapes["l_cell"][s]), ra)
        assert np.array_equal(obs[e], o["obs"].T)
        for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
            assert np.array_equal(idx[k][e].cpu().numpy(), o[k]), (e, k)
    assert (np.abs(p) <= 3.4 + 1e-12).all() and (np.abs(dp) <= 0.5).all()
    # a different episode / seed gives a different draw; the same one reproduces
    obs2 = sb.reset(seed=226, episode=4, env_offset=env_offset).cpu().numpy()
    assert not np.array_equal(obs, obs2)
    obs3 = sb.reset(seed=226, episode=3, env_offset=env_offset).cpu().numpy()
    assert np.array_equal(obs, obs3)
    sb.close()


def test_sharded_reset_equals_global_reset(shapes):
    """Two 'ranks' resetting env slices [0,8) and [8,16) reproduce the single 16-env reset (no communication)."""
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.shapes import r_avoid_for
    n_a = 32
    ra = r_avoid_for(n_a, shapes)
    ng_max = max(np.asarray(g).shape[0] for g in shapes["grid_coords"])
    full = SwarmBatch(n_env=16, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra); full.set_shapes(shapes)
    o_full = full.reset(seed=9, episode=0).clone()
    parts = []
    for r in range(2):
        sb = SwarmBatch(n_env=8, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra); sb.set_shapes(shapes)
        parts.append(sb.reset(seed=9, episode=0, env_offset=8 * r).clone())
        sb.close()
    assert torch.equal(o_full, torch.cat(parts, dim=0))
    full.close()


def test_device_rollout_and_replay(shapes):
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.rollout import DeviceReplay, PolicyMLP, rollout
    from marl_llm_amd.shapes import r_avoid_for
    E, N = 64, 32
    ra = r_avoid_for(N, shapes)
    ng_max = max(np.asarray(g).shape[0] for g in shapes["grid_coords"])
    sb = SwarmBatch(n_env=E, n_agents=N, n_cells_max=ng_max, r_avoid=ra); sb.set_shapes(shapes)
    obs = sb.reset(seed=1)
    torch.manual_seed(0)
    policy = PolicyMLP(obs_dim=sb.obs_dim).to(sb.device)
    replay = DeviceReplay(capacity_rows=5 * E * N + 17, obs_dim=sb.obs_dim, act_dim=2, device=sb.device)
    first = obs.clone()
    obs, rews = rollout(sb, policy, steps=4, obs=obs, replay=replay, noise_scale=0.1)
    assert len(replay) == 4 * E * N and rews.shape == (4,)
    n = E * N
    assert torch.equal(replay.obs[:n], first.reshape(n, -1))
    for t in range(3):      # consecutive transitions chain: next_obs[t] == obs[t+1]
        assert torch.equal(replay.next_obs[t * n:(t + 1) * n], replay.obs[(t + 1) * n:(t + 2) * n])
    assert torch.equal(replay.next_obs[3 * n:4 * n], obs.reshape(n, -1))
    assert replay.act.abs().max() <= 1 and set(replay.rew.unique().tolist()).issubset({0.0, 1.0})
    # ring behaviour of buffer_agent.py:97-100: a block that would overflow is written flush with the end
    obs, _ = rollout(sb, policy, steps=2, obs=obs, replay=replay)
    assert len(replay) == 6 * n and replay.curr_i == 0      # flush-with-the-end write; the fill counter overshoots like the reference's
    b = replay.sample(512)
    assert b[0].shape == (512, sb.obs_dim) and b[1].shape == (512, 2)
    sb.close()


def test_chained_replay_holds_the_same_transitions(shapes):
    """ChainedReplay (one observation block per step) against DeviceReplay (two) on the same rollout: identical
    transitions, also after the ring wrapped; sampled (obs, next_obs) pairs are consecutive steps of the same row."""
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.rollout import ChainedReplay, DeviceReplay, PolicyMLP, rollout
    from marl_llm_amd.shapes import r_avoid_for
    E, N, K = 16, 32, 4
    n = E * N
    ng_max = max(np.asarray(g).shape[0] for g in shapes["grid_coords"])
    sb = SwarmBatch(n_env=E, n_agents=N, n_cells_max=ng_max, r_avoid=r_avoid_for(N, shapes)); sb.set_shapes(shapes)
    obs = sb.reset(seed=3)
    torch.manual_seed(0)
    policy = PolicyMLP(obs_dim=sb.obs_dim).to(sb.device)
    flat, chain = DeviceReplay(K * n, sb.obs_dim, 2, sb.device), ChainedReplay(K, n, sb.obs_dim, 2, sb.device)

    class Both:
        def push(self, *a):
            flat.push(*a); chain.push(*a)
    obs, _ = rollout(sb, policy, steps=7, obs=obs, replay=Both(), noise_scale=0.1)       # 7 > K: both rings wrapped
    assert len(chain) == K * n == len(flat)
    # flat buffer: rows [curr_i, ...) oldest first; chain: the K steps before slot `cur`
    order = [(flat.curr_i // n + k) % K for k in range(K)]                              # flat block index of step (newest - K + 1 + k)
    for k, fb in enumerate(order):
        j = (chain.cur - K + k) % chain.S
        sl = slice(fb * n, (fb + 1) * n)
        assert torch.equal(chain.obs[j], flat.obs[sl]) and torch.equal(chain.obs[(j + 1) % chain.S], flat.next_obs[sl])
        assert torch.equal(chain.act[j], flat.act[sl]) and torch.equal(chain.rew[j], flat.rew[sl])
        assert torch.equal(chain.act_prior[j], flat.act_prior[sl])
    assert torch.equal(chain.obs[chain.cur], obs.reshape(n, -1))                          # the newest next_obs
    g = torch.Generator(device=sb.device).manual_seed(1)
    o, a, r, no, d, pr = chain.sample(256, generator=g)
    assert o.shape == (256, sb.obs_dim) and no.shape == o.shape and a.shape == (256, 2)
    sb.close()


@pytest.mark.parametrize("n_a,n_env", [(8, 5), (30, 6), (64, 8), (200, 3), (256, 2)])
def test_eval_metrics_match_wrapper_restatement(shapes, n_a, n_env):
    """coverage / min-distance uniformity / Voronoi uniformity (assembly_wrapper.py:48-128) from the kernel equal the
    reference's Python loops restated in oracle_py.wrapper_metrics (numpy pairwise summation reproduced): exact."""
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.shapes import r_avoid_for
    from oracle.oracle_py import wrapper_metrics
    ra = r_avoid_for(n_a, shapes)
    ng_max = max(np.asarray(g).shape[0] for g in shapes["grid_coords"])
    sb = SwarmBatch(n_env=n_env, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra); sb.set_shapes(shapes)
    sb.reset(seed=4)
    act = torch.zeros((n_env, n_a, 2), device=sb.device)
    for _ in range(30):                      # let the prior policy pull the swarm onto the shape
        act = sb.step(act)[3]
    m = sb.metrics().cpu().numpy()
    p, _ = [x.cpu().numpy() for x in sb.get_state()]
    cells, n_g = sb.get_cells()
    for e in range(n_env):
        ref = wrapper_metrics(p[e], np.ascontiguousarray(cells[e][:, : n_g[e]]), ra)
        assert np.array_equal(m[e], ref, equal_nan=True), (e, m[e], ref)
    assert (m[:, 0] > 0).any()
    sb.close()
