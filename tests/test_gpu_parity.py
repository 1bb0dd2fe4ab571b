"""GPU parity tests: the HIP path (through the C ABI, via marl_llm_amd.batched.SwarmBatch) against the oracle
(oracle/assembly_oracle.c) and against the golden vectors recorded from the reference.

Tolerances, stated once:
  * state (p, dp), every index / flag array, done and reward: BIT-EXACT (==).
  * obs / a_prior with obs_dtype=float64: BIT-EXACT.
  * obs / a_prior with obs_dtype=float32 (the product dtype): equal to the oracle's double value rounded once to
    float32 -- i.e. exact equality after casting the oracle's output to float32 (|err| <= 2^-24 relative).
  The only outputs that touch a non-correctly-rounded primitive are the reward's cos() terms (device libm vs
  glibc, <= a few ulp); a flip of the 0.05 threshold from that has probability ~1e-14 per agent-step.
"""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, golden_files, load_golden, make_case

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _batch(**kw):
    from marl_llm_amd.batched import SwarmBatch
    return SwarmBatch(**kw)


def _to_rows(obs_cols):
    """oracle obs (D, N) -> rows (N, D)."""
    return np.ascontiguousarray(obs_cols.T)


def _pad_cells(grids, ng_max):
    E = len(grids)
    cells = np.zeros((E, 2, ng_max))
    n_g = np.zeros(E, np.int32)
    for e, g in enumerate(grids):
        n_g[e] = g.shape[1]
        cells[e, :, : g.shape[1]] = g
    return cells, n_g


@pytest.mark.parametrize("path", golden_files(), ids=os.path.basename)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_golden_steps(path, dtype):
    """Every recorded reference step, teacher-forced through the HIP path (E = 1)."""
    z = load_golden(path)
    T, _, n_a = z["p"].shape
    n_g = z["grid"].shape[1]
    odt = torch.float64 if dtype == "f64" else torch.float32
    sb = _batch(n_env=1, n_agents=n_a, n_cells_max=n_g, r_avoid=float(z["r_avoid"]), d_sen=float(z["d_sen"]),
                is_boundary=bool(z["is_boundary"]), with_self=bool(z["with_self"]), obs_dtype=odt,
                boundary=tuple(z["boundary"]))
    sb.set_cells(z["grid"][None], [n_g], [float(z["l_cell"])])
    for t in range(T):
        sb.set_state(z["p"][t][None], z["dp"][t][None])
        sb.observe()
        assert np.array_equal(sb.indices(False, False)["neighbor_index"][0].cpu().numpy(), z["nei_prev"][t])
        act = torch.from_numpy(np.ascontiguousarray(z["a"][t].T)[None]).to(sb.device)       # (2,N) -> [1,N,2]
        obs, rew, done, pri = sb.step(act)
        p, dp = sb.get_state()
        assert np.array_equal(p[0].cpu().numpy(), z["p_next"][t])
        assert np.array_equal(dp[0].cpu().numpy(), z["dp_next"][t])
        idx = sb.indices()
        assert np.array_equal(idx["neighbor_index"][0].cpu().numpy(), z["nei"][t])
        assert np.array_equal(idx["in_flags"][0].cpu().numpy(), z["in_flags"][t])
        assert np.array_equal(idx["sensed_index"][0].cpu().numpy(), z["sensed"][t])
        assert np.array_equal(idx["occupied_index"][0].cpu().numpy(), z["occupied"][t])
        assert np.array_equal(rew[0].cpu().numpy().astype(np.float64), z["rew"][t][0])
        assert not done.any().item()
        ref_obs = _to_rows(z["obs"][t]); ref_pri = _to_rows(z["a_prior"][t])
        if dtype == "f32":
            ref_obs = ref_obs.astype(np.float32); ref_pri = ref_pri.astype(np.float32)
        assert np.array_equal(obs[0].cpu().numpy(), ref_obs)
        assert np.array_equal(pri[0].cpu().numpy(), ref_pri)
    sb.close()


@pytest.mark.parametrize("n_a", [30, 100, 200])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_golden_batches(n_a, dtype):
    """Three recorded reference episodes of one agent count (own seed, own target shape: ragged cell sets) teacher-forced
    through the HIP path as ONE 3-env batch: the batched layout, env-indexed cell sets and the agent counts that are not
    powers of two held against the reference itself (make_golden.py r3b), not only against the pinned oracle."""
    zs = [load_golden(p) for p in golden_files(f"g11_n{n_a}_s*.npz")]
    assert len(zs) == 3
    E, T = len(zs), min(z["p"].shape[0] for z in zs)
    cells, n_g = _pad_cells([z["grid"] for z in zs], max(z["grid"].shape[1] for z in zs))
    assert len({float(z["r_avoid"]) for z in zs}) == 1 and len(set(n_g.tolist())) > 1
    odt = torch.float64 if dtype == "f64" else torch.float32
    sb = _batch(n_env=E, n_agents=n_a, n_cells_max=cells.shape[2], r_avoid=float(zs[0]["r_avoid"]), d_sen=float(zs[0]["d_sen"]),
                obs_dtype=odt, boundary=tuple(zs[0]["boundary"]))
    sb.set_cells(cells, n_g, [float(z["l_cell"]) for z in zs])
    st = lambda k, t: np.stack([z[k][t] for z in zs])
    for t in range(T):
        sb.set_state(st("p", t), st("dp", t))
        sb.observe()
        assert np.array_equal(sb.indices(False, False)["neighbor_index"].cpu().numpy(), st("nei_prev", t))
        act = torch.from_numpy(np.ascontiguousarray(st("a", t).transpose(0, 2, 1))).to(sb.device)      # (E,2,N) -> [E,N,2]
        obs, rew, done, pri = sb.step(act)
        p, dp = sb.get_state()
        assert np.array_equal(p.cpu().numpy(), st("p_next", t)) and np.array_equal(dp.cpu().numpy(), st("dp_next", t))
        idx = sb.indices()
        for k_dev, k_ref in (("neighbor_index", "nei"), ("in_flags", "in_flags"), ("sensed_index", "sensed"), ("occupied_index", "occupied")):
            assert np.array_equal(idx[k_dev].cpu().numpy(), st(k_ref, t)), (k_dev, t)
        assert np.array_equal(rew.cpu().numpy().astype(np.float64), st("rew", t)[:, 0])
        assert not done.any().item()
        ref_obs = np.ascontiguousarray(st("obs", t).transpose(0, 2, 1)); ref_pri = np.ascontiguousarray(st("a_prior", t).transpose(0, 2, 1))
        if dtype == "f32":
            ref_obs = ref_obs.astype(np.float32); ref_pri = ref_pri.astype(np.float32)
        assert np.array_equal(obs.cpu().numpy(), ref_obs)
        assert np.array_equal(pri.cpu().numpy(), ref_pri)
    sb.close()


def test_known_answer_case():
    """SURVEY.md section 8c hand-checkable case (topo=2, G=4, OCC=5) through the HIP path."""
    z = load_golden(os.path.join(GOLDEN_DIR, "g1_kat_n3.npz"))
    sb = _batch(n_env=1, n_agents=3, n_cells_max=4, r_avoid=0.15, d_sen=0.4, topo=2, g_max=4, occ_max=5,
                obs_dtype=torch.float64)
    sb.set_cells(z["grid"][None], [4], [0.06])
    sb.set_state(z["p"][None], z["dp"][None])
    obs = sb.observe()
    idx = sb.indices()
    assert idx["neighbor_index"][0].cpu().tolist() == [[1, -1], [0, -1], [-1, -1]]
    assert idx["in_flags"][0].cpu().tolist() == [1, 0, 0]
    assert idx["sensed_index"][0].cpu().tolist() == [[2, -1, -1, -1], [0, 1, 2, -1], [-1, -1, -1, -1]]
    assert idx["occupied_index"][0].cpu().tolist() == [[0, 1, -1, -1, -1], [-1] * 5, [-1] * 5]
    assert np.array_equal(obs[0].cpu().numpy(), _to_rows(z["obs"]))
    sb.close()


CONFIGS = [  # (n_a, n_env, cluster, periodic, with_self, steps)
    (3, 5, 1, False, True, 4), (8, 16, 1, False, True, 6), (8, 9, 0, True, True, 4), (30, 6, 1, False, True, 6),
    (32, 8, 1, False, False, 6), (32, 5, 0, False, True, 4), (64, 6, 1, False, True, 8), (64, 4, 0, False, True, 4),
    (64, 3, 1, True, True, 4), (100, 3, 1, False, True, 3), (128, 2, 1, False, True, 3), (256, 2, 1, False, True, 3),
    (200, 2, 0, True, False, 2),
]


@pytest.mark.parametrize("flags", [0, 2], ids=["lattice", "generic"])
@pytest.mark.parametrize("n_a,n_env,cluster,periodic,with_self,steps", CONFIGS)
def test_batched_trajectories_vs_oracle(oracle, shapes, n_a, n_env, cluster, periodic, with_self, steps, flags):
    """E independent envs with different shapes / rotations, free-running for several steps: the state
    trajectory, all masks and fp64 outputs stay bit-identical to E sequential oracle envs."""
    from marl_llm_amd.shapes import r_avoid_for
    rng = np.random.default_rng(7000 + 13 * n_a + n_env)
    ra = r_avoid_for(n_a, shapes)
    cases = [make_case(rng, shapes, n_a, cluster) for _ in range(n_env)]
    ng_max = max(c[2].shape[1] for c in cases) + 3
    cells, n_g = _pad_cells([c[2] for c in cases], ng_max)
    sb = _batch(n_env=n_env, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra, is_boundary=not periodic,
                with_self=with_self, obs_dtype=torch.float64, debug_flags=flags)
    sb.set_cells(cells, n_g, [c[3] for c in cases])
    # the synthetic shapes are tiled lattices like the reference's: recognised unless the path is disabled
    assert sb.lattice_envs() == (n_env if flags == 0 else 0)
    p = np.stack([c[0] for c in cases]); dp = np.stack([c[1] for c in cases])
    sb.set_state(p, dp)
    obs0 = sb.observe().cpu().numpy()
    idx = sb.indices()
    nei = []
    for e, (pe, dpe, g, l_cell) in enumerate(cases):
        o = oracle.get_observation(pe, dpe, g, l_cell, ra, is_periodic=periodic, with_self=with_self)
        assert np.array_equal(obs0[e], _to_rows(o["obs"])), e
        for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
            assert np.array_equal(idx[k][e].cpu().numpy(), o[k]), (e, k)
        nei.append(o["neighbor_index"])
    state = [(c[0], c[1]) for c in cases]
    for t in range(steps):
        if t % 2 == 0:
            act = rng.uniform(-1, 1, (n_env, n_a, 2)).astype(np.float32)
        else:   # feed the prior back (assembles the swarm; exercises the occupied filter harder)
            act = last_prior.astype(np.float32)
        obs, rew, done, pri = sb.step(torch.from_numpy(act).to(sb.device))
        obs, rew, pri = obs.cpu().numpy(), rew.cpu().numpy(), pri.cpu().numpy()
        pg, dpg = [x.cpu().numpy() for x in sb.get_state()]
        idx = sb.indices()
        for e in range(n_env):
            s = oracle.step(state[e][0], state[e][1], np.ascontiguousarray(act[e].T), cases[e][2], nei[e], cases[e][3],
                            ra, is_boundary=not periodic, with_self=with_self)
            assert np.array_equal(pg[e], s["p"]) and np.array_equal(dpg[e], s["dp"]), (t, e)
            assert np.array_equal(obs[e], _to_rows(s["obs"])), (t, e)
            assert np.array_equal(pri[e], _to_rows(s["a_prior"])), (t, e)
            assert np.array_equal(rew[e].astype(np.float64), s["reward"][0]), (t, e)
            for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
                assert np.array_equal(idx[k][e].cpu().numpy(), s[k]), (t, e, k)
            state[e] = (s["p"], s["dp"]); nei[e] = s["neighbor_index"]
        last_prior = pri
        assert not done.any().item()
    sb.close()


def test_small_caps(oracle, shapes):
    """Shrunk caps so both round(i*step) sub-samplings (80-cell and 200-cell lists) are exercised."""
    from marl_llm_amd.shapes import r_avoid_for
    rng = np.random.default_rng(5)
    n_a, n_env = 32, 4
    ra = r_avoid_for(n_a, shapes)
    cases = [make_case(rng, shapes, n_a, 1) for _ in range(n_env)]
    ng_max = max(c[2].shape[1] for c in cases)
    cells, n_g = _pad_cells([c[2] for c in cases], ng_max)
    # (G-1 odd -> integer cap arithmetic; G-1 even -> the reference's fp64 round(), ties possible)
    for topo, g_max, occ_max in ((3, 10, 7), (6, 80, 20), (1, 6, 200), (2, 5, 9), (6, 81, 33), (4, 21, 11)):
        sb = _batch(n_env=n_env, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra, topo=topo, g_max=g_max,
                    occ_max=occ_max, obs_dtype=torch.float64)
        sb.set_cells(cells, n_g, [c[3] for c in cases])
        sb.set_state(np.stack([c[0] for c in cases]), np.stack([c[1] for c in cases]))
        obs = sb.observe().cpu().numpy()
        idx = sb.indices()
        for e, (pe, dpe, g, l_cell) in enumerate(cases):
            o = oracle.get_observation(pe, dpe, g, l_cell, ra, topo=topo, g_max=g_max, occ_max=occ_max)
            assert np.array_equal(obs[e], _to_rows(o["obs"]))
            for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
                assert np.array_equal(idx[k][e].cpu().numpy(), o[k]), (e, k)
        sb.close()


@pytest.mark.parametrize("n_a,n_env,env_offset", [
    (64, 4096, 0),            # BASELINE config 2 (headline): one wavefront per env and split
    (32, 1024, 0),            # BASELINE config 1: two envs per wavefront (EPB = 2)
    (256, 4096, 0),           # BASELINE config 4: 1024-thread workgroups, LDS-atomic pair masks, dense O(N^2) path
    (64, 4096, 3 * 4096),     # one rank's shard of BASELINE config 3 (64 x 32768 over 8 GPUs): envs [12288, 16384)
    (32, 1023, 0),            # env count not divisible by the envs per workgroup: the `e >= n_env` tail lanes
    (30, 1021, 0),            # the reference's default agent count, padded lanes inside every wavefront
], ids=["64x4096", "32x1024", "256x4096", "64x4096_shard3", "32x1023", "30x1021"])
def test_full_size_properties(oracle, shapes, n_a, n_env, env_offset):
    """Every BASELINE shape at FULL size: size-independent invariants over the whole batch plus an exact oracle
    comparison on a sample of environments (incl. the first and the last), in the product dtype (float32 obs)."""
    from marl_llm_amd.shapes import r_avoid_for
    from marl_llm_amd.synth import synthetic_batch
    ra = r_avoid_for(n_a, shapes)
    sy = synthetic_batch(n_env, n_a, shapes, seed=226, assembled_fraction=0.5, env_offset=env_offset)
    if env_offset:            # a shard is the same slice of the global generation (counter-based inputs)
        ref = synthetic_batch(8, n_a, shapes, seed=226, assembled_fraction=0.5, env_offset=env_offset + 100)
        assert np.array_equal(ref["p"], sy["p"][100:108]) and np.array_equal(ref["cells"], sy["cells"][100:108])
    sb = _batch(n_env=n_env, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=ra)
    sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"])
    sb.set_state(sy["p"], sy["dp"])
    sb.observe()
    nei0 = sb.indices(False, False)["neighbor_index"].cpu().numpy()
    rng = np.random.default_rng(1)
    act = rng.uniform(-1, 1, (n_env, n_a, 2)).astype(np.float32)
    obs, rew, done, pri = sb.step(torch.from_numpy(act).to(sb.device))
    obs_c, rew_c, pri_c = obs.cpu().numpy(), rew.cpu().numpy(), pri.cpu().numpy()
    pg, dpg = [x.cpu().numpy() for x in sb.get_state()]
    assert np.isfinite(obs_c).all() and np.isfinite(pg).all()
    assert not done.any().item()
    assert set(np.unique(rew_c)).issubset({0.0, 1.0})
    assert np.abs(dpg).max() <= 0.8 and np.abs(pri_c).max() <= 1.0
    assert (obs_c[:, :, 0] == pg[:, 0, :].astype(np.float32)).all()          # self block = absolute state
    idx = sb.indices()
    sen = idx["sensed_index"].cpu().numpy()
    pad = sen < 0                                                            # unused slots are zero in obs
    sl = obs_c[:, :, 32:].reshape(n_env, n_a, 80, 2)
    assert (sl[pad] == 0).all()
    assert ((sen[:, :, 1:] < 0) | (sen[:, :, :-1] >= 0)).all()               # valid slots form a prefix
    assert (np.diff(np.where(sen >= 0, sen, 1 << 20), axis=-1) > 0)[(sen[:, :, 1:] >= 0)].all()   # ascending cell index
    assert sb.lattice_envs() == n_env                                       # tiled shapes: the row-walk path (N <= 64)
    sample = np.unique(np.concatenate([[0, 1, n_env - 2, n_env - 1], rng.choice(n_env, 24 if n_a <= 64 else 10, replace=False)]))
    for e in sample:
        g = sy["cells"][e][:, : sy["n_g"][e]]
        s = oracle.step(sy["p"][e], sy["dp"][e], np.ascontiguousarray(act[e].T), g, nei0[e], float(sy["l_cell"][e]), ra)
        assert np.array_equal(pg[e], s["p"]) and np.array_equal(dpg[e], s["dp"])
        assert np.array_equal(obs_c[e], _to_rows(s["obs"]).astype(np.float32))
        assert np.array_equal(pri_c[e], _to_rows(s["a_prior"]).astype(np.float32))
        assert np.array_equal(rew_c[e].astype(np.float64), s["reward"][0])
        for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
            assert np.array_equal(idx[k][e].cpu().numpy(), s[k]), (e, k)
    sb.close()


def test_error_behaviour(shapes):
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd._lib import SwarmError
    with pytest.raises(SwarmError):
        SwarmBatch(n_env=1, n_agents=300, n_cells_max=10, r_avoid=0.1)
    sb = SwarmBatch(n_env=2, n_agents=8, n_cells_max=600, r_avoid=0.5)
    with pytest.raises(SwarmError):                  # step before cells/state/observe
        sb.step(torch.zeros((2, 8, 2), device=sb.device))
    with pytest.raises(SwarmError):
        sb.observe()
    with pytest.raises(SwarmError):                  # wrong action shape
        sb.step(torch.zeros((2, 7, 2), device=sb.device))
    sb.close()


def _adversarial_case(rng, shapes, n_a, ra, d_sen=0.4):
    """Agents placed (almost) exactly ON the decision thresholds the fp32 pre-filter has to resolve:
    distance to a cell ~ d_sen and ~ r_avoid/2 (sensed / occupied bits), the midpoint between two cells
    (nearest-cell ties), the in-shape radius, and agent pairs ~ d_sen, r_avoid and 0.07 apart."""
    p, dp, g, l_cell = make_case(rng, shapes, n_a, 1)
    eps = [0.0, 1e-16, -1e-16, 1e-13, -1e-13, 1e-10, -1e-10, 1e-8, -1e-8, 3e-7, -3e-7, 2e-6, -2e-6]
    k = 0
    for i in range(n_a):
        c = int(rng.integers(0, g.shape[1]))
        th = rng.uniform(0, 2 * np.pi)
        u = np.array([np.cos(th), np.sin(th)])
        mode = i % 6
        e = eps[k % len(eps)]; k += 1
        if mode == 0:
            p[:, i] = g[:, c] + u * d_sen * (1 + e)
        elif mode == 1:
            p[:, i] = g[:, c] + u * (ra / 2) * (1 + e)
        elif mode == 2:
            c2 = (c + 1) % g.shape[1]
            mid = 0.5 * (g[:, c] + g[:, c2]); d = g[:, c2] - g[:, c]
            p[:, i] = mid + d * e + np.array([-d[1], d[0]]) * rng.uniform(-0.3, 0.3)
        elif mode == 3:
            p[:, i] = g[:, c] + u * (np.sqrt(2) * l_cell / 2) * (1 + e)
        elif mode == 4 and i > 0:
            p[:, i] = p[:, i - 1] + u * [d_sen, ra, 0.07, d_sen + ra / 2][k % 4] * (1 + e)
        elif mode == 5:
            # nearest-cell ties across lattice rows: the midpoint of two vertically adjacent cells, the common vertex
            # of a 2x2 block (four-way tie -> lowest index wins), and the same from far outside the shape
            dd = np.linalg.norm(g - g[:, [c]], axis=0)
            nb = np.where((dd > 0) & (dd < 1.01 * l_cell))[0]
            if len(nb):
                d = g[:, nb[int(rng.integers(0, len(nb)))]] - g[:, c]
                perp = np.array([-d[1], d[0]])
                sub = k % 3
                if sub == 0:
                    p[:, i] = g[:, c] + 0.5 * d + d * e
                elif sub == 1:
                    p[:, i] = g[:, c] + 0.5 * d + 0.5 * perp + u * abs(e)
                else:
                    p[:, i] = g[:, c] + 0.5 * d + perp * (rng.integers(3, 12) + 0.5) + d * e
    return np.ascontiguousarray(p), dp, g, l_cell


@pytest.mark.parametrize("n_a,n_env,force", [(64, 24, 0), (64, 8, 1), (64, 12, 2), (64, 6, 3), (32, 16, 0), (32, 8, 2), (8, 16, 0),
                                             (256, 3, 0), (100, 4, 1)])
def test_threshold_adversarial_inputs(oracle, shapes, n_a, n_env, force):
    """The fp32 pre-filter must hand every borderline decision to the exact fp64 path: masks, flags and the
    step stay bit-identical to the oracle on inputs constructed to sit on the thresholds.  force=1 runs the
    same inputs with every exact fallback forced (debug flag bit 0); bit 1 disables the lattice row walk, so both
    the lattice and the generic all-cells paths are exercised -- all must agree with the oracle."""
    from marl_llm_amd.shapes import r_avoid_for
    rng = np.random.default_rng(4242 + n_a + force)
    ra = r_avoid_for(n_a, shapes)
    cases = [_adversarial_case(rng, shapes, n_a, ra) for _ in range(n_env)]
    ng_max = max(c[2].shape[1] for c in cases)
    cells, n_g = _pad_cells([c[2] for c in cases], ng_max)
    sb = _batch(n_env=n_env, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra, obs_dtype=torch.float64, debug_flags=force)
    sb.set_cells(cells, n_g, [c[3] for c in cases])
    sb.set_state(np.stack([c[0] for c in cases]), np.stack([c[1] for c in cases]))
    obs0 = sb.observe().cpu().numpy()
    idx = sb.indices()
    nei = []
    for e, (pe, dpe, g, l_cell) in enumerate(cases):
        o = oracle.get_observation(pe, dpe, g, l_cell, ra)
        for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
            assert np.array_equal(idx[k][e].cpu().numpy(), o[k]), (e, k)
        assert np.array_equal(obs0[e], _to_rows(o["obs"])), e
        nei.append(o["neighbor_index"])
    act = np.zeros((n_env, n_a, 2), np.float32)          # zero action: the agents stay near the thresholds
    obs, rew, done, pri = sb.step(torch.from_numpy(act).to(sb.device))
    idx = sb.indices()
    pg, dpg = [x.cpu().numpy() for x in sb.get_state()]
    for e, (pe, dpe, g, l_cell) in enumerate(cases):
        s = oracle.step(pe, dpe, np.ascontiguousarray(act[e].T), g, nei[e], l_cell, ra)
        assert np.array_equal(pg[e], s["p"]) and np.array_equal(dpg[e], s["dp"])
        assert np.array_equal(obs[e].cpu().numpy(), _to_rows(s["obs"]))
        assert np.array_equal(rew[e].cpu().numpy().astype(np.float64), s["reward"][0])
        for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
            assert np.array_equal(idx[k][e].cpu().numpy(), s[k]), (e, k)
    sb.close()


def test_forced_exact_paths_equal_fast_paths(shapes):
    """Whole-batch consistency: fast and forced-exact runs, lattice walk and generic scan (debug_flags 0..3) give
    identical outputs over several free-running steps at a BASELINE-sized agent count (tools/stress_consistency.py
    runs the same comparison at millions of agent-steps)."""
    from marl_llm_amd.shapes import r_avoid_for
    from marl_llm_amd.synth import synthetic_batch
    n_a, n_env = 64, 512
    ra = r_avoid_for(n_a, shapes)
    sy = synthetic_batch(n_env, n_a, shapes, seed=11, assembled_fraction=0.6)
    outs = []
    for force in (0, 1, 2, 3):
        sb = _batch(n_env=n_env, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=ra, debug_flags=force)
        sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"]); sb.set_state(sy["p"], sy["dp"]); sb.observe()
        act = torch.zeros((n_env, n_a, 2), device=sb.device)
        rews = []
        for t in range(12):
            obs, rew, done, act = sb.step(act)
            rews.append(rew.clone())
        p, dp = sb.get_state()
        idx = sb.indices()
        outs.append((obs.clone(), torch.stack(rews), p, dp, idx["sensed_index"], idx["occupied_index"], idx["in_flags"]))
        sb.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert torch.equal(a, b)
    assert outs[0][1].sum().item() > 0          # some agents do earn the reward in this workload


def test_diagnostic_repeat_hooks_do_not_change_results(shapes):
    """tools/ablate.py relies on every phase being idempotent: running any phase extra times (debug hook) must
    leave every output bit-identical."""
    from marl_llm_amd.shapes import r_avoid_for
    from marl_llm_amd.synth import synthetic_batch
    n_a, n_env = 64, 96
    ra = r_avoid_for(n_a, shapes)
    sy = synthetic_batch(n_env, n_a, shapes, seed=5, assembled_fraction=0.6)
    ref = None
    for flags in [0] + [(k << 8) | (2 << 12) for k in range(1, 13)]:
        sb = _batch(n_env=n_env, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=ra, debug_flags=flags)
        sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"]); sb.set_state(sy["p"], sy["dp"]); sb.observe()
        act = torch.zeros((n_env, n_a, 2), device=sb.device)
        for t in range(4):
            obs, rew, done, act = sb.step(act)
        p, dp = sb.get_state()
        idx = sb.indices()
        out = (obs.clone(), rew.clone(), act.clone(), p, dp, idx["sensed_index"], idx["occupied_index"], idx["neighbor_index"])
        sb.close()
        if ref is None:
            ref = out
        else:
            for a, b in zip(ref, out):
                assert torch.equal(a, b), flags


def test_non_lattice_cells_fall_back_to_generic_path(oracle, shapes):
    """Arbitrary (jittered, shuffled) cell sets are not a lattice: the generic scan serves them, same results."""
    from marl_llm_amd.shapes import r_avoid_for
    rng = np.random.default_rng(77)
    n_a, n_env = 32, 6
    ra = r_avoid_for(n_a, shapes)
    cases = []
    for k in range(n_env):
        p, dp, g, l_cell = make_case(rng, shapes, n_a, 1)
        if k % 2 == 0:
            g = g + rng.normal(0, 0.004, g.shape)               # jitter: off-lattice
        else:
            g = np.ascontiguousarray(g[:, rng.permutation(g.shape[1])])   # on-lattice points, but not row-major order
        cases.append((p, dp, np.ascontiguousarray(g), l_cell))
    ng_max = max(c[2].shape[1] for c in cases)
    cells, n_g = _pad_cells([c[2] for c in cases], ng_max)
    sb = _batch(n_env=n_env, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra, obs_dtype=torch.float64)
    sb.set_cells(cells, n_g, [c[3] for c in cases])
    assert sb.lattice_envs() == 0
    sb.set_state(np.stack([c[0] for c in cases]), np.stack([c[1] for c in cases]))
    obs0 = sb.observe().cpu().numpy()
    idx = sb.indices()
    for e, (pe, dpe, g, l_cell) in enumerate(cases):
        o = oracle.get_observation(pe, dpe, g, l_cell, ra)
        assert np.array_equal(obs0[e], _to_rows(o["obs"]))
        for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
            assert np.array_equal(idx[k][e].cpu().numpy(), o[k]), (e, k)
    sb.close()


@pytest.mark.parametrize("n_a", [64, 24])
def test_wide_lattice_uses_64_bit_row_masks(oracle, n_a):
    """A shape wider than 32 lattice columns (the reference's are not) takes the 64-bit row-mask instantiation of the
    lattice walk: observation, index scratch and a step against the oracle, including agents on the decision thresholds."""
    rng = np.random.default_rng(9 + n_a)
    l_cell, ra = 0.055, 0.12
    cols, rows = 44, 12
    keep = rng.uniform(size=(rows, cols)) > 0.12                       # holes; every row keeps some cells
    keep[:, 0] = True; keep[:, -1] = True
    bb, aa = np.nonzero(keep)                                          # row-major: rows ascending, columns ascending
    base = np.stack([aa * l_cell, bb * l_cell]).astype(np.float64)
    n_env = 6
    cases = []
    for e in range(n_env):
        th = rng.uniform(-np.pi, np.pi)
        rot = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
        g = np.ascontiguousarray(rot @ (base - base.mean(1, keepdims=True)) + rng.uniform(-0.8, 0.8, (2, 1)))
        p = g[:, rng.integers(0, g.shape[1], n_a)] + rng.normal(0, 0.05, (2, n_a))
        for i in range(0, n_a, 3):                                     # threshold cases: d_sen, r_avoid/2, midpoints
            c = int(rng.integers(0, g.shape[1])); u = rng.normal(size=2); u /= np.linalg.norm(u)
            p[:, i] = g[:, c] + u * [0.4, ra / 2, 0.5 * l_cell][i % 3] * (1 + [0.0, 1e-13, -1e-10][(i // 3) % 3])
        dp = rng.uniform(-0.3, 0.3, (2, n_a))
        cases.append((np.ascontiguousarray(p), dp, g))
    ng = cases[0][2].shape[1]
    cells, n_g = _pad_cells([c[2] for c in cases], ng)
    sb = _batch(n_env=n_env, n_agents=n_a, n_cells_max=ng, r_avoid=ra, obs_dtype=torch.float64)
    sb.set_cells(cells, n_g, [l_cell] * n_env)
    assert sb.lattice_envs() == n_env
    sb.set_state(np.stack([c[0] for c in cases]), np.stack([c[1] for c in cases]))
    obs0 = sb.observe().cpu().numpy()
    idx = sb.indices()
    nei = []
    for e, (pe, dpe, g) in enumerate(cases):
        o = oracle.get_observation(pe, dpe, g, l_cell, ra)
        for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
            assert np.array_equal(idx[k][e].cpu().numpy(), o[k]), (e, k)
        assert np.array_equal(obs0[e], _to_rows(o["obs"])), e
        nei.append(o["neighbor_index"])
    act = rng.uniform(-1, 1, (n_env, n_a, 2)).astype(np.float32)
    obs, rew, done, pri = sb.step(torch.from_numpy(act).to(sb.device))
    idx = sb.indices()
    for e, (pe, dpe, g) in enumerate(cases):
        s = oracle.step(pe, dpe, np.ascontiguousarray(act[e].T.astype(np.float64)), g, nei[e], l_cell, ra)
        assert np.array_equal(obs[e].cpu().numpy(), _to_rows(s["obs"]))
        assert np.array_equal(rew[e].cpu().numpy().astype(np.float64), s["reward"][0])
        for k in ("neighbor_index", "in_flags", "sensed_index", "occupied_index"):
            assert np.array_equal(idx[k][e].cpu().numpy(), s[k]), (e, k)
    sb.close()


@pytest.mark.parametrize("n_a,periodic", [(64, False), (36, True), (200, False)])
def test_exact_distance_ties_in_the_neighbour_list(oracle, shapes, n_a, periodic):
    """Agents on an exactly representable square grid: every agent has several neighbours at EXACTLY equal distance, so the
    order of its list is decided by the tie rule alone (lower index first -- the reference's std::sort leaves ties
    unspecified, CPP:641; oracle and kernel both take the lower index).  Exercises the exact fallback of the key-based
    neighbour selection (keys that agree in all but the index bits)."""
    from marl_llm_amd.shapes import r_avoid_for
    rng = np.random.default_rng(4)
    ra = r_avoid_for(n_a, shapes)
    side = int(np.ceil(np.sqrt(n_a)))
    E = 3
    cases = []
    for e in range(E):
        _, dp, g, l_cell = make_case(rng, shapes, n_a, 0)
        pitch = [0.125, 0.1875, 0.25][e]                    # exact in binary: distances tie exactly
        ij = rng.permutation(side * side)[:n_a]            # random index <-> grid position assignment
        p = np.stack([(ij % side) * pitch - 1.0, (ij // side) * pitch - 1.0])
        cases.append((np.ascontiguousarray(p), dp, g, l_cell))
    ng_max = max(c[2].shape[1] for c in cases)
    cells, n_g = _pad_cells([c[2] for c in cases], ng_max)
    sb = _batch(n_env=E, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra, is_boundary=not periodic, obs_dtype=torch.float64)
    sb.set_cells(cells, n_g, [c[3] for c in cases])
    sb.set_state(np.stack([c[0] for c in cases]), np.stack([c[1] for c in cases]))
    obs = sb.observe().cpu().numpy()
    idx = sb.indices()
    ties = 0
    for e, (p, dp, g, l_cell) in enumerate(cases):
        o = oracle.get_observation(p, dp, g, l_cell, ra, is_periodic=periodic)
        assert np.array_equal(idx["neighbor_index"][e].cpu().numpy(), o["neighbor_index"]), e
        assert np.array_equal(obs[e], _to_rows(o["obs"])), e
        nei = o["neighbor_index"]
        for i in range(n_a):                                # count lists that really contain a tie
            js = nei[i][nei[i] >= 0]
            d = np.sum((p[:, js] - p[:, [i]]) ** 2, axis=0)
            ties += int(len(d) > 1 and (np.diff(d) == 0).any())
    assert ties > n_a                                       # most lists do
    # one free step: rewards / priors on tied lists (collision flag = nearest listed neighbour)
    act = torch.zeros((E, n_a, 2), dtype=torch.float64, device=sb.device)
    obs2, rew, done, pri = sb.step(act)
    for e, (p, dp, g, l_cell) in enumerate(cases):
        o = oracle.get_observation(p, dp, g, l_cell, ra, is_periodic=periodic)
        s = oracle.step(p, dp, np.zeros((2, n_a)), g, o["neighbor_index"], l_cell, ra, is_boundary=not periodic)
        assert np.array_equal(obs2[e].cpu().numpy(), _to_rows(s["obs"])), e
        assert np.array_equal(rew[e].cpu().numpy().astype(np.float64), s["reward"][0]), e
        assert np.array_equal(pri[e].cpu().numpy(), _to_rows(s["a_prior"])), e
    sb.close()


def test_near_ties_whose_norms_coincide(oracle, shapes):
    """Two neighbours whose SQUARED distances differ by an ulp or two but whose norms (sqrt) are equal: the reference
    sorts by the norm (CPP:636-641), so they tie and the lower index comes first -- although its squared distance is the
    LARGER one.  A selection that orders by squared distance gets these lists wrong."""
    from marl_llm_amd.shapes import r_avoid_for
    n_a, E = 8, 6
    rng = np.random.default_rng(11)
    ra = r_avoid_for(n_a, shapes)
    cases, collapsed = [], 0
    while len(cases) < E:
        x, y = rng.uniform(0.05, 0.25, 2)
        hit = None
        for k in range(1, 6):
            bx = x
            for _ in range(k):
                bx = np.nextafter(bx, 1.0)
            d_small, d_large = x * x + y * y, y * y + bx * bx
            if d_small != d_large and np.sqrt(d_small) == np.sqrt(d_large):
                hit = bx
                break
        if hit is None:
            continue
        _, dp, g, l_cell = make_case(rng, shapes, n_a, 0)
        p = rng.uniform(-2.3, 2.3, (2, n_a))
        p[:, 0] = 0.0                                      # agent 0 at the origin: relative positions are exact
        p[:, 1] = (y, hit)                                 # lower index, larger squared distance, same norm
        p[:, 2] = (x, y)
        far = np.sum(p[:, 3:] ** 2, axis=0) < 0.45 ** 2    # keep the others out of agent 0's sensing range
        p[:, 3:][:, far] += 1.0
        cases.append((np.ascontiguousarray(p), dp, g, l_cell))
    ng_max = max(c[2].shape[1] for c in cases)
    cells, n_g = _pad_cells([c[2] for c in cases], ng_max)
    sb = _batch(n_env=E, n_agents=n_a, n_cells_max=ng_max, r_avoid=ra, obs_dtype=torch.float64)
    sb.set_cells(cells, n_g, [c[3] for c in cases])
    sb.set_state(np.stack([c[0] for c in cases]), np.stack([c[1] for c in cases]))
    obs = sb.observe().cpu().numpy()
    idx = sb.indices()
    for e, (p, dp, g, l_cell) in enumerate(cases):
        o = oracle.get_observation(p, dp, g, l_cell, ra)
        assert list(o["neighbor_index"][0][:2]) == [1, 2]           # the oracle (= reference rule): tie, lower index first
        d2 = np.sum(p[:, 1:3] ** 2, axis=0)
        collapsed += int(d2[0] > d2[1])
        assert np.array_equal(idx["neighbor_index"][e].cpu().numpy(), o["neighbor_index"]), e
        assert np.array_equal(obs[e], _to_rows(o["obs"])), e
    assert collapsed == E
    sb.close()


@pytest.mark.parametrize("n_a,n_env,periodic", [(8, 37, False), (16, 9, True), (30, 21, False), (32, 64, False), (32, 5, True)])
def test_half_occupied_geometry_equals_the_full_one(oracle, shapes, n_a, n_env, periodic):
    """Small batches of N < 64 agents run with half of the agent threads empty and eight lanes per agent in the list phase
    (twice the workgroups: Geo<NPAD, true>); debug_flags bit 2 keeps the full geometry.  Both must give the same bits as
    each other on free-running steps, and the oracle's on the last one."""
    from marl_llm_amd.shapes import r_avoid_for
    from marl_llm_amd.synth import synthetic_batch
    ra = r_avoid_for(n_a, shapes)
    sy = synthetic_batch(n_env, n_a, shapes, seed=17 + n_a, assembled_fraction=0.7)
    outs = []
    for flags in (0, 4):
        sb = _batch(n_env=n_env, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=ra, is_boundary=not periodic,
                    obs_dtype=torch.float64, debug_flags=flags)
        sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"]); sb.set_state(sy["p"], sy["dp"]); sb.observe()
        act = torch.zeros((n_env, n_a, 2), dtype=torch.float64, device=sb.device)
        rews = []
        for t in range(12):
            if t == 11:
                p0, dp0 = [x.cpu().numpy() for x in sb.get_state()]
                nei0 = sb.indices(False, False)["neighbor_index"].cpu().numpy()
                a0 = act.cpu().numpy()
            obs, rew, done, pri = sb.step(act)
            act = pri.clone()
            rews.append(rew.clone())
        idx = sb.indices()
        p, dp = sb.get_state()
        outs.append((obs.clone(), torch.stack(rews), pri.clone(), p, dp, idx["sensed_index"], idx["occupied_index"],
                     idx["neighbor_index"], idx["in_flags"], p0, dp0, nei0, a0))
        sb.close()
    for a, b in zip(outs[0][:9], outs[1][:9]):
        assert torch.equal(a, b)
    obs, rews, pri, p, dp, sen, occ, nei, inf, p0, dp0, nei0, a0 = outs[0]
    for e in range(min(n_env, 6)):
        g = sy["cells"][e][:, : sy["n_g"][e]]
        s = oracle.step(p0[e], dp0[e], np.ascontiguousarray(a0[e].T), g, nei0[e], float(sy["l_cell"][e]), ra, is_boundary=not periodic)
        assert np.array_equal(p[e].cpu().numpy(), s["p"]) and np.array_equal(dp[e].cpu().numpy(), s["dp"])
        assert np.array_equal(obs[e].cpu().numpy(), _to_rows(s["obs"]))
        assert np.array_equal(rews[-1][e].cpu().numpy().astype(np.float64), s["reward"][0])
        assert np.array_equal(pri[e].cpu().numpy(), _to_rows(s["a_prior"]))
        assert np.array_equal(sen[e].cpu().numpy(), s["sensed_index"]) and np.array_equal(occ[e].cpu().numpy(), s["occupied_index"])
