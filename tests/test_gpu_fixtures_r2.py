"""Round-2 pins against outputs of the reference itself (tests/golden/make_golden.py r2):
  * the reference's own target shapes (fig/*.png -> tests/golden/fig_cells.npz) through the batched HIP path: lattice
    path taken, bit-exact vs the oracle; the recorded reference episodes on those shapes (g6_*) are covered by the
    parametrised golden tests in test_gpu_parity.py / test_oracle_golden.py;
  * the device-side reset (k_reset) against statistics of 10^4 reference reset() calls (g7);
  * the rollout's actor and replay buffer against the reference's MLPNetwork outputs (g8) and ReplayBufferAgent
    contents (g9)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, fig_shapes, load_golden

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def test_real_shapes_take_the_lattice_path_and_match_the_oracle(oracle):
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.shapes import r_avoid_for
    from marl_llm_amd.synth import synthetic_batch
    fig = fig_shapes()
    for n_a in (64, 32):
        E = 56
        ra = r_avoid_for(n_a, fig)
        sy = synthetic_batch(E, n_a, fig, seed=11, assembled_fraction=0.6)
        assert len(set(sy["n_g"].tolist())) >= 5                               # several of the seven shapes are present
        sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=ra)
        sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"]); sb.set_state(sy["p"], sy["dp"]); sb.observe()
        assert sb.lattice_envs() == E                                          # every real shape is a lattice subset
        p, dp = sy["p"].copy(), sy["dp"].copy()
        nei = sb.indices(False, False)["neighbor_index"].cpu().numpy()
        act = torch.zeros((E, n_a, 2), device=sb.device)
        for t in range(3):
            obs, rew, done, pri = sb.step(act)
            a = act.cpu().numpy()
            idx = sb.indices()
            pg, dpg = [x.cpu().numpy() for x in sb.get_state()]
            for e in range(E):
                g = sy["cells"][e][:, : sy["n_g"][e]]
                s = oracle.step(p[e], dp[e], np.ascontiguousarray(a[e].T.astype(np.float64)), g, nei[e], float(sy["l_cell"][e]), ra)
                assert np.array_equal(pg[e], s["p"]) and np.array_equal(dpg[e], s["dp"]), (t, e)
                assert np.array_equal(obs[e].cpu().numpy(), s["obs"].T.astype(np.float32)), (t, e)
                assert np.array_equal(pri[e].cpu().numpy(), s["a_prior"].T.astype(np.float32)), (t, e)
                assert np.array_equal(rew[e].cpu().numpy().astype(np.float64), s["reward"][0]), (t, e)
                for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
                    assert np.array_equal(idx[k][e].cpu().numpy(), s[k]), (t, e, k)
                p[e], dp[e], nei[e] = s["p"], s["dp"], s["neighbor_index"]
            act = pri.clone()
        sb.close()
    # the device-side reset on the real shape set: still lattices after rotation + offset
    sb = SwarmBatch(n_env=64, n_agents=32, n_cells_max=max(g.shape[0] for g in fig["grid_coords"]), r_avoid=r_avoid_for(32, fig))
    sb.set_shapes(fig)
    sb.reset(seed=5)
    assert sb.lattice_envs() == 64 and set(sb.get_shape_index().tolist()) == set(range(7))
    sb.close()


def test_device_reset_distribution_matches_the_reference(shapes):
    """k_reset against 10^4 reset() calls of the reference (g7, N = 8, synthetic shape set): shape histogram, the 50/50
    arena / 2x2-box branch, ranges of positions, velocities and offsets, rotation angle, and the quantiles of every
    drawn quantity -- distribution equality within sampling error of two 10^4-sized samples."""
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.shapes import r_avoid_for
    z = load_golden(os.path.join(GOLDEN_DIR, "g7_reset_stats_n8.npz"))
    R, N = int(z["n_resets"]), int(z["n_agents"])
    ng_max = max(np.asarray(g).shape[0] for g in shapes["grid_coords"])
    sb = SwarmBatch(n_env=R, n_agents=N, n_cells_max=ng_max, r_avoid=r_avoid_for(N, shapes))
    sb.set_shapes(shapes)
    sb.reset(seed=12345)
    P, DP = [x.cpu().numpy() for x in sb.get_state()]
    cells, n_g = sb.get_cells()
    SH = sb.get_shape_index()
    S = len(shapes["l_cell"])
    hist = np.bincount(SH, minlength=S)
    sig = np.sqrt(R / S * (1 - 1 / S))
    assert (np.abs(hist - R / S) < 5 * sig).all() and (np.abs(z["shape_hist"] - R / S) < 5 * sig).all()
    span = P.max(axis=2) - P.min(axis=2)
    cluster = (span <= 2.0).all(axis=1)
    assert abs(cluster.sum() - R / 2) < 5 * np.sqrt(R) / 2 and abs(int(z["n_cluster"]) - R / 2) < 5 * np.sqrt(R) / 2
    # ranges: inside the same supports as the reference's draws (assembly.py:184-185,203-208,215)
    assert -3.4 <= P.min() and P.max() <= 3.4 and abs(P.min() - z["p_min"]) < 0.1 and abs(P.max() - z["p_max"]) < 0.1
    assert -0.5 <= DP.min() and DP.max() <= 0.5 and abs(DP.min() - z["dp_min"]) < 1e-3 and abs(DP.max() - z["dp_max"]) < 1e-3
    OFF = np.stack([np.array([cells[e][0, : n_g[e]].mean(), cells[e][1, : n_g[e]].mean()]) for e in range(R)])
    assert (OFF.min(axis=0) >= -1.4 - 1e-9).all() and (OFF.max(axis=0) <= 1.4 + 1e-9).all()
    assert np.abs(OFF.min(axis=0) - z["off_min"]).max() < 0.01 and np.abs(OFF.max(axis=0) - z["off_max"]).max() < 0.01
    ANG = np.empty(R)
    for e in range(R):
        o = np.asarray(shapes["grid_coords"][SH[e]], np.float64)[0]
        v = cells[e][:, 0] - OFF[e]
        ANG[e] = np.arctan2(o[0] * v[1] - o[1] * v[0], o[0] * v[0] + o[1] * v[1])
    q = z["quantiles"]

    def close(sample, ref_q, scale):
        """two-sample quantile comparison: every reference quantile within `4 / sqrt(n)` of the sample's CDF"""
        s = np.sort(np.asarray(sample).reshape(-1))
        cdf = np.searchsorted(s, ref_q, side="right") / s.size
        assert np.abs(cdf - q)[1:-1].max() < 4.0 / np.sqrt(min(s.size, scale)), (np.abs(cdf - q).max())

    close(ANG, z["ang_quant"], R)
    close(OFF[:, 0], z["off_quant"][:, 0], R); close(OFF[:, 1], z["off_quant"][:, 1], R)
    close(P[~cluster], z["p_spread_quant"], R)
    close(DP, z["dp_quant"], R)
    close((P[cluster].max(axis=2) + P[cluster].min(axis=2)) / 2, z["cluster_centre_quant"], R)
    close(P[cluster] - P[cluster].mean(axis=2, keepdims=True), z["cluster_rel_quant"], R)
    sb.close()


def _actor_from_fixture(z):
    from marl_llm_amd.rollout import PolicyMLP
    m = PolicyMLP(obs_dim=192, act_dim=2, hidden_dim=180)
    sd = {f"fc{k}.{n}": torch.from_numpy(z[f"fc{k}_{n}"]) for k in (1, 2, 3, 4) for n in ("weight", "bias")}
    m.load_state_dict(sd)
    return m


def test_actor_against_the_reference_mlpnetwork():
    """The reference's MLPNetwork(192, 2, hidden 180, tanh out) on fixed weights / inputs (networks.py:6-44, recorded on
    the CPU in fp32).  PolicyMLP (torch, fp32) must reproduce it to GEMM rounding; FusedPolicy -- the rollout's actor -- is
    a bf16 kernel (bf16 weights and activations, fp32 accumulation) and is held to the bf16 tolerance stated here."""
    from marl_llm_amd.rollout import FusedPolicy
    z = load_golden(os.path.join(GOLDEN_DIR, "g8_mlp_actor.npz"))
    m = _actor_from_fixture(z).to("cuda:0")
    x = torch.from_numpy(z["X"]).to("cuda:0")
    with torch.no_grad():
        y32 = m(x).cpu().numpy()
    assert np.abs(y32 - z["Y"]).max() <= 2e-5
    fused = FusedPolicy(m, device="cuda")                    # device without an index: normalised to cuda:N
    y16 = fused(x.contiguous()).cpu().numpy()
    err = np.abs(y16 - z["Y"])
    assert err.max() <= 4e-2 and err.mean() <= 6e-3, (err.max(), err.mean())
    y16b = fused(x.to(torch.bfloat16).contiguous()).cpu().numpy()            # bf16 observation rows (rollout mode)
    assert np.abs(y16b - z["Y"]).max() <= 6e-2
    fused.close()
    # the accurate actor mode: operands split into high + low bf16 parts, three MFMAs per product -- the rollout follows the
    # reference's fp32 MLPNetwork itself, not a bf16 approximation of it
    x3 = FusedPolicy(m, device="cuda", precision="bf16x3")
    y3 = x3(x.contiguous()).cpu().numpy()
    e3 = np.abs(y3 - z["Y"])
    assert e3.max() <= 1e-4 and e3.mean() <= 2e-5, (e3.max(), e3.mean())
    assert np.abs(y3 - y32).max() <= 1e-4
    # exploration noise and the in-place output work in this mode too
    out = torch.empty_like(torch.from_numpy(y3)).to("cuda:0")
    y3n = x3(x.contiguous(), out=out, noise_scale=0.1, seed=3, step=9)
    assert y3n.data_ptr() == out.data_ptr() and (y3n.cpu().numpy() != y3).any() and y3n.abs().max() <= 1
    x3.close()


def test_replay_against_the_reference_buffer():
    """ReplayBufferAgent.push (buffer_agent.py:67-128) replayed block by block, incl. the overflow rule (:97-100: a block
    that would run past the end is written flush with it) and the fill counter that overshoots the capacity (:122-123):
    identical write positions, counters and contents."""
    from marl_llm_amd.rollout import DeviceReplay
    z = load_golden(os.path.join(GOLDEN_DIR, "g9_replay_push.npz"))
    dev = torch.device("cuda:0")
    rb = DeviceReplay(capacity_rows=z["obs"].shape[0], obs_dim=4, act_dim=2, device=dev, obs_dtype=torch.float64)
    rb.act, rb.act_prior, rb.rew, rb.done = [t.to(torch.float64) for t in (rb.act, rb.act_prior, rb.rew, rb.done)]
    for t, n in enumerate(z["blocks"]):
        n = int(n)
        f = lambda k: torch.from_numpy(np.ascontiguousarray(z["in_" + k][t][:, :n].T)).to(dev)      # (dim, agents) -> rows
        rb.push(f("o").reshape(n, 1, -1), f("a").reshape(n, 1, -1), f("r").reshape(n, 1), f("o2").reshape(n, 1, -1),
                f("d").reshape(n, 1), f("ap").reshape(n, 1, -1))
        assert (rb.curr_i, rb.filled_i, len(rb)) == tuple(int(v) for v in z["log"][t]), t
    for name, ref in (("obs", "obs"), ("act", "act"), ("rew", "rew"), ("next_obs", "next_obs"), ("done", "done"),
                      ("act_prior", "act_prior")):
        assert np.array_equal(getattr(rb, name).cpu().numpy(), z[ref]), name
    # sampling: the default draws only written rows; the reference's sliding window needs its 3e5-row head room
    o = rb.sample(64)[0]
    assert o.shape == (64, 4)
    with pytest.raises(ValueError):
        rb.sample_reference(16)
    big = DeviceReplay(capacity_rows=300000 + 5000, obs_dim=2, act_dim=2, device=dev)
    big.obs[:, 0] = torch.arange(big.capacity, device=dev, dtype=torch.float32)
    g = torch.Generator(device=dev); g.manual_seed(3)
    rows = big.sample_reference(512, generator=g)[0][:, 0].cpu().numpy().astype(np.int64)
    assert len(set(rows.tolist())) == 512 and rows.max() - rows.min() < 5000          # distinct rows of one 5000-row window
