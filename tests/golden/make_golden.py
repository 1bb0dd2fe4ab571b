#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference and oracle/_ref/libAssemblyEnv.so, which
oracle/Makefile compiles from the reference's own C++).  It imports the reference's Python env
(cus_gym/gym/envs/customized_envs/assembly.py) from where it lies and records inputs / outputs as data;
nothing of the reference's source is copied.  The committed *.npz files are what travels to the GPU box.

The reference's loader (envs_cplus/c_lib.py:11-22) looks for ``build/libAssemblyEnv.so`` beside itself in
the read-only reference tree, so ``ctypes.CDLL`` is redirected to the library oracle/Makefile built from
the same sources.  ``results.pkl`` (a missing large blob in the reference) is replaced by the synthetic
shape set of marl_llm_amd/shapes.py written in the reference's pickle layout to a temp dir.

Usage:  MPLBACKEND=Agg python tests/golden/make_golden.py        (round-1 fixtures g1..g5)
        MPLBACKEND=Agg python tests/golden/make_golden.py r2     (round-2 fixtures: real shapes, thicker g2, reset
                                                                  statistics, learner-side modules)
        MPLBACKEND=Agg python tests/golden/make_golden.py r3     (round-3 fixtures: agent_strategy == 'llm')
        MPLBACKEND=Agg python tests/golden/make_golden.py r3b    (round-3 fixtures: three episodes each of N = 30 / 100 / 200,
                                                                  run by the tests as one 3-env batch)
"""
import ctypes
import os
import sys
import tempfile
import types

os.environ.setdefault("MPLBACKEND", "Agg")
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.dont_write_bytecode = True

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libAssemblyEnv.so")
REF_PY = "/root/reference/cus_gym"

from marl_llm_amd.shapes import save_results, synthetic_shape_set  # noqa: E402
from oracle.oracle_py import RefLib  # noqa: E402


def import_reference_env():
    orig = ctypes.CDLL

    class Redirect(orig):
        def __init__(self, name, *a, **k):
            if isinstance(name, str) and name.endswith("libAssemblyEnv.so"):
                name = REF_SO
            super().__init__(name, *a, **k)

    ctypes.CDLL = Redirect
    sys.path.insert(0, REF_PY)
    import gym  # the reference's vendored gym 0.19 fork
    from gym.wrappers import AssemblySwarmWrapper
    import gym.envs.customized_envs  # noqa: F401  (assembly.py loads the library at import, assembly.py:13)
    ctypes.CDLL = orig
    return gym, AssemblySwarmWrapper


def make_env(gym, Wrapper, n_a, pkl, is_boundary=True, with_self=True, strategy="input", collected=False):
    args = types.SimpleNamespace(n_a=n_a, render_traj=False, traj_len=15, is_collected=collected, video=False,
                                 is_boundary=is_boundary, dynamics_mode="Cartesian", agent_strategy=strategy,
                                 is_con_self_state=with_self, is_feature_norm=False, training_method="llm_rl",
                                 results_file=pkl)
    return Wrapper(gym.make("AssemblySwarm-v0").unwrapped, args)


FIELDS = ("p", "dp", "a", "nei_prev", "p_next", "dp_next", "obs", "rew", "done", "a_prior",
          "nei", "in_flags", "sensed", "occupied")


def record_episode(env, n_steps, mode, rng, warm=0):
    """Free-running episode; every step is recorded with its full pre-state so that it is both a
    teacher-forced single-step vector (G2) and part of a trajectory (G3)."""
    base = env.env
    obs = env.reset()
    n_a = base.n_a
    a = np.zeros((2, n_a), np.float32)
    for _ in range(warm):                          # assemble the swarm: action := prior (assembly.py:663-666)
        _, _, _, _, ap = env.step(a)
        a = ap.astype(np.float32)
    rec = {k: [] for k in FIELDS}
    for _ in range(n_steps):
        if mode == "random":
            a = rng.uniform(-1, 1, (2, n_a)).astype(np.float32)
        rec["p"].append(base.p.copy()); rec["dp"].append(base.dp.copy()); rec["a"].append(a.copy())
        rec["nei_prev"].append(base.neighbor_index.copy())
        o, r, d, _, ap = env.step(a)
        rec["p_next"].append(base.p.copy()); rec["dp_next"].append(base.dp.copy())
        rec["obs"].append(o.copy()); rec["rew"].append(r.copy()); rec["done"].append(d.copy())
        rec["a_prior"].append(ap.copy()); rec["nei"].append(base.neighbor_index.copy())
        rec["in_flags"].append(base.in_flags.copy()); rec["sensed"].append(base.sensed_index.copy())
        rec["occupied"].append(base.occupied_index.copy())
        if mode == "prior":
            a = ap.astype(np.float32)
    out = {k: np.stack(v) for k, v in rec.items()}
    # the wrapper's evaluation metrics on the final state (assembly_wrapper.py:48-128)
    with np.errstate(all="ignore"):
        out["metrics"] = np.array([env.coverage_rate(), env.distribution_uniformity(), env.voronoi_based_uniformity()])
    out.update(grid=base.grid_center.copy(), l_cell=np.float64(base.l_cell), r_avoid=np.float64(base.r_avoid),
               d_sen=np.float64(base.d_sen), boundary=base.boundary_pos.copy(),
               is_boundary=np.bool_(base.is_boundary), with_self=np.bool_(base.is_con_self_state))
    return out


def main():
    gym, Wrapper = import_reference_env()
    tmp = tempfile.mkdtemp(prefix="golden_")
    pkl = os.path.join(tmp, "results.pkl")
    save_results(pkl, synthetic_shape_set())
    rng = np.random.default_rng(226)

    # ---- G1: the hand-checkable known-answer case of SURVEY.md section 8c, straight from the reference .so
    ref = RefLib()
    p = np.array([[0, 0.1, 1], [0, 0, 1]], np.float64); dp = np.array([[0.1, 0.2, 0.3], [0, -0.1, 0.5]], np.float64)
    cells = np.array([[0, 0.05, 0.3, 1.5], [0.02, 0, 0, 1.5]], np.float64)
    o = ref.get_observation(p, dp, cells, 0.06, 0.15, d_sen=0.4, topo=2, g_max=4, occ_max=5)
    rew = ref.get_reward(p, cells, o["neighbor_index"], o["in_flags"], o["sensed_index"], 0.15, d_sen=0.4,
                         occupied_index=o["occupied_index"])
    prior = ref.action_prior(p, dp, cells, o["neighbor_index"], 0.06, 0.15, d_sen=0.4)
    np.savez_compressed(os.path.join(HERE, "g1_kat_n3.npz"), p=p, dp=dp, grid=cells, l_cell=0.06, r_avoid=0.15,
                        d_sen=0.4, topo=2, g_max=4, occ_max=5, rew=rew, a_prior=prior, **o)

    # ---- G2/G3: recorded episodes of the reference's Python env
    plan = [(8, "random", 0, 4, True, True), (8, "prior", 60, 4, True, True),
            (8, "random", 0, 3, False, True),            # periodic boundary
            (8, "prior", 60, 3, True, False),            # is_con_self_state = False
            (32, "random", 0, 3, True, True), (32, "prior", 100, 3, True, True),
            (64, "random", 0, 3, True, True), (64, "prior", 100, 3, True, True),
            (256, "prior", 100, 1, True, True)]
    for n_a, mode, warm, steps, is_boundary, with_self in plan:
        np.random.seed(226 + n_a)                 # the env draws from the global numpy RNG (assembly.py:156-215)
        env = make_env(gym, Wrapper, n_a, pkl, is_boundary, with_self)
        rec = record_episode(env, steps, mode, rng, warm)
        tag = f"g2_n{n_a}_{mode}" + ("" if is_boundary else "_periodic") + ("" if with_self else "_noself")
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
        print(tag, "in_shape", rec["in_flags"].sum(1), "reward", rec["rew"].sum((1, 2)),
              "occupied", (rec["occupied"] >= 0).sum((1, 2)), "sensed max", (rec["sensed"] >= 0).sum(2).max())

    # ---- G5: the rule-based expert (agent_strategy='rule', is_collected=True: step returns u, assembly.py:530-601,663-664)
    for n_a in (8, 32):
        np.random.seed(500 + n_a)
        env = make_env(gym, Wrapper, n_a, pkl, strategy="rule", collected=True)
        env.reset()
        b = env.env
        rec = {k: [] for k in ("p", "dp", "u", "p_next", "dp_next", "rew")}
        for t in range(40):
            pre_p, pre_dp = b.p.copy(), b.dp.copy()
            o, r, d, _, u = env.step(np.zeros((2, n_a), np.float32))          # the passed action is ignored in rule mode
            if t >= 34:                                                          # keep the last steps (swarm partly assembled)
                rec["p"].append(pre_p); rec["dp"].append(pre_dp); rec["u"].append(u.copy())
                rec["p_next"].append(b.p.copy()); rec["dp_next"].append(b.dp.copy()); rec["rew"].append(r.copy())
        out = {k: np.stack(v) for k, v in rec.items()}
        out.update(grid=b.grid_center.copy(), l_cell=np.float64(b.l_cell), r_avoid=np.float64(b.r_avoid), d_sen=np.float64(b.d_sen))
        np.savez_compressed(os.path.join(HERE, f"g5_rule_n{n_a}.npz"), **out)
        print("g5_rule", n_a, "|u| max", np.abs(out["u"]).max(), "reward", out["rew"].sum((1, 2)))

    # ---- G4: reset() draw order under seed 226 (assembly_cfg.py:174 default seed), N = 8
    np.random.seed(226)
    env = make_env(gym, Wrapper, 8, pkl)          # __reinit__ consumes n_a^2 draws (assembly.py:133)
    obs0 = env.reset()
    b = env.env
    np.savez_compressed(os.path.join(HERE, "g4_reset_seed226_n8.npz"), p=b.p, dp=b.dp, grid=b.grid_center,
                        l_cell=np.float64(b.l_cell), r_avoid=np.float64(b.r_avoid), obs=obs0,
                        shape_frequency=np.asarray(b.shape_frequency))
    print("done ->", HERE)


def main_r2():
    """Round-2 fixtures.  (a) the reference's own target shapes fig/*.png, tiled by marl_llm_amd.shape_images (the
    reference's cv2 pipeline cannot run here: cv2 is not installed), fed to the reference env; (b) thicker g2 coverage:
    3 steps at N = 256, periodic at N = 32 / 64, is_con_self_state = False at N = 64; (c) statistics of 10^4 reference
    reset() calls; (d) outputs of the reference's MLPNetwork / ReplayBufferAgent (learner side of the device rollout)."""
    from marl_llm_amd.shape_images import pack_cells_npz, process_folder
    gym, Wrapper = import_reference_env()
    tmp = tempfile.mkdtemp(prefix="golden_r2_")
    rng = np.random.default_rng(2262)

    # ---- (a) real shapes
    fig = process_folder("/root/reference/fig")
    pack_cells_npz(fig, os.path.join(HERE, "fig_cells.npz"))
    pkl_fig = os.path.join(tmp, "results_fig.pkl")
    save_results(pkl_fig, fig)
    for n_a, mode, warm, seed in ((32, "prior", 100, 1), (32, "random", 0, 2), (64, "prior", 100, 3), (64, "prior", 100, 4),
                                  (64, "random", 0, 5)):
        np.random.seed(1000 + seed)
        env = make_env(gym, Wrapper, n_a, pkl_fig)
        rec = record_episode(env, 3, mode, rng, warm)
        rec["shape_index"] = np.int64(int(np.argmax(env.env.shape_frequency)))
        tag = f"g6_fig_n{n_a}_{mode}_s{seed}"
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
        print(tag, "shape", rec["shape_index"], "n_g", rec["grid"].shape[1], "in_shape", rec["in_flags"].sum(1),
              "reward", rec["rew"].sum((1, 2)), "sensed max", (rec["sensed"] >= 0).sum(2).max())

    # ---- (b) thicker g2 (synthetic shape set, as round 1)
    pkl = os.path.join(tmp, "results.pkl")
    save_results(pkl, synthetic_shape_set())
    for n_a, mode, warm, steps, is_boundary, with_self, tag in (
            (256, "prior", 100, 3, True, True, "g2_n256_prior3"),
            (32, "random", 0, 3, False, True, "g2_n32_random_periodic"),
            (64, "prior", 100, 3, False, True, "g2_n64_prior_periodic"),
            (64, "prior", 100, 3, True, False, "g2_n64_prior_noself")):
        np.random.seed(7000 + n_a)
        env = make_env(gym, Wrapper, n_a, pkl, is_boundary, with_self)
        rec = record_episode(env, steps, mode, rng, warm)
        np.savez_compressed(os.path.join(HERE, tag + ".npz"), **rec)
        print(tag, "in_shape", rec["in_flags"].sum(1), "reward", rec["rew"].sum((1, 2)))

    # ---- (c) reset() statistics over 10^4 calls of the reference (assembly.py:156-219), N = 8, synthetic shape set
    np.random.seed(31337)
    env = make_env(gym, Wrapper, 8, pkl)
    b = env.env
    R = 10000
    origins = [g.T for g in b.grid_center_origins] if hasattr(b, "grid_center_origins") else None
    P = np.empty((R, 2, 8)); DP = np.empty((R, 2, 8)); OFF = np.empty((R, 2)); ANG = np.empty(R); SH = np.empty(R, np.int64)
    prev = np.asarray(b.shape_frequency, np.float64).copy()
    for k in range(R):
        env.reset()
        P[k] = b.p; DP[k] = b.dp
        now = np.asarray(b.shape_frequency, np.float64)
        SH[k] = int(np.argmax(now - prev)); prev = now.copy()
        g = b.grid_center
        OFF[k] = g.mean(axis=1)                              # the shape frame is centred, so the cells' mean is the offset
        o = np.asarray(b.grid_center_origins[SH[k]], np.float64).T if origins is not None else None
        # rotation angle from the first cell: g0 - offset = R(angle) o0
        v = g[:, 0] - OFF[k]; u = o[:, 0]
        ANG[k] = np.arctan2(u[0] * v[1] - u[1] * v[0], u[0] * v[0] + u[1] * v[1])
    span = P.max(axis=2) - P.min(axis=2)
    cluster = (span <= 2.0).all(axis=1)                       # the 2 x 2 box branch (:207-208); the arena branch spans more
    q = np.linspace(0.0, 1.0, 21)
    np.savez_compressed(os.path.join(HERE, "g7_reset_stats_n8.npz"), n_resets=np.int64(R), n_agents=np.int64(8),
                        shape_hist=np.bincount(SH, minlength=len(b.l_cells)), n_cluster=np.int64(cluster.sum()),
                        p_min=P.min(), p_max=P.max(), dp_min=DP.min(), dp_max=DP.max(),
                        off_min=OFF.min(axis=0), off_max=OFF.max(axis=0), off_quant=np.quantile(OFF, q, axis=0),
                        ang_quant=np.quantile(ANG, q), ang_min=ANG.min(), ang_max=ANG.max(),
                        p_spread_quant=np.quantile(P[~cluster].reshape(-1), q), dp_quant=np.quantile(DP.reshape(-1), q),
                        cluster_centre_quant=np.quantile(((P[cluster].max(axis=2) + P[cluster].min(axis=2)) / 2).reshape(-1), q),
                        cluster_rel_quant=np.quantile((P[cluster] - P[cluster].mean(axis=2, keepdims=True)).reshape(-1), q),
                        quantiles=q)
    print("g7 reset stats: shapes", np.bincount(SH, minlength=len(b.l_cells)), "cluster", cluster.sum(), "of", R)

    # ---- (d) learner-side modules the device rollout mirrors: MLPNetwork (networks.py:6-44), ReplayBufferAgent
    #      (buffer_agent.py:13-128).  Imported from where they lie; only arrays are stored.
    import importlib.util
    import torch

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod

    nets = load("ref_networks", "/root/reference/marl_llm/algorithm/utils/networks.py")
    buf = load("ref_buffer_agent", "/root/reference/marl_llm/algorithm/utils/buffer_agent.py")
    torch.manual_seed(226)
    net = nets.MLPNetwork(192, 2, hidden_dim=180, constrain_out=True)          # maddpg.py / agents.py:23-27 actor shape
    with torch.no_grad():
        for prm in net.parameters():                                           # spread the outputs over tanh's range
            prm.mul_(3.0)
    with np.load(os.path.join(HERE, "g2_n64_prior.npz")) as z:
        real = z["obs"][0].T.astype(np.float32)                                # 64 real observation rows
    X = np.concatenate([real, rng.normal(0, 0.5, (192, 192)).astype(np.float32)], axis=0)
    with torch.no_grad():
        Y = net(torch.from_numpy(X)).numpy()
    sd = {k: v.numpy() for k, v in net.state_dict().items()}
    np.savez_compressed(os.path.join(HERE, "g8_mlp_actor.npz"), X=X, Y=Y, **{k.replace(".", "_"): v for k, v in sd.items()})
    print("g8 mlp: |Y| max", np.abs(Y).max(), "mean", np.abs(Y).mean())

    # replay: 5 steps x 6 agents = 30 rows; blocks of 6, 6, 6, 6, 4 rows, then 6 (overflow rule :97-100), then 4, 6
    rb = buf.ReplayBufferAgent(5, 6, slice(0, 6), state_dim=4, action_dim=2)
    log = []
    blocks = (6, 6, 6, 6, 4, 6, 4, 6)
    ins = []
    for t, n in enumerate(blocks):
        o = rng.normal(size=(4, 6)); a = rng.normal(size=(2, 6)); r = rng.normal(size=(1, 6)); o2 = rng.normal(size=(4, 6))
        d = (rng.uniform(size=(1, 6)) < 0.3); ap = rng.normal(size=(2, 6))
        rb.push(o, a, r, o2, d, slice(0, n), ap)
        ins.append(dict(o=o, a=a, r=r, o2=o2, d=d.astype(np.float64), ap=ap))
        log.append((rb.curr_i, rb.filled_i, len(rb)))
    np.savez_compressed(os.path.join(HERE, "g9_replay_push.npz"), blocks=np.array(blocks), log=np.array(log),
                        obs=rb.obs_buffs, act=rb.ac_buffs, rew=rb.rew_buffs, next_obs=rb.next_obs_buffs, done=rb.done_buffs,
                        act_prior=rb.ac_prior_buffs,
                        **{f"in_{k}": np.stack([x[k] for x in ins]) for k in ("o", "a", "r", "o2", "d", "ap")})
    print("g9 replay log (curr_i, filled_i, len):", log)
    print("done ->", HERE)


def main_r3():
    """Round-3 fixtures: agent_strategy == 'llm' (assembly.py:525-529: the agents are driven by robot_prior_policy, the
    Python twin of the prior with repulsion gain 1.0, assembly.py:892-940).  is_collected makes step() return the applied
    action u (assembly.py:663-664)."""
    gym, Wrapper = import_reference_env()
    tmp = tempfile.mkdtemp(prefix="golden_")
    pkl = os.path.join(tmp, "results.pkl")
    save_results(pkl, synthetic_shape_set())
    for n_a, keep_from, steps in ((8, 50, 56), (32, 70, 76)):
        np.random.seed(700 + n_a)
        env = make_env(gym, Wrapper, n_a, pkl, strategy="llm", collected=True)
        env.reset()
        b = env.env
        rec = {k: [] for k in ("p", "dp", "nei_prev", "u", "p_next", "dp_next", "rew", "obs")}
        for t in range(steps):
            pre_p, pre_dp, pre_nei = b.p.copy(), b.dp.copy(), b.neighbor_index.copy()
            o, r, d, _, u = env.step(np.zeros((2, n_a), np.float32))          # the passed action is ignored in llm mode
            if t >= keep_from:
                rec["p"].append(pre_p); rec["dp"].append(pre_dp); rec["nei_prev"].append(pre_nei); rec["u"].append(u.copy())
                rec["p_next"].append(b.p.copy()); rec["dp_next"].append(b.dp.copy()); rec["rew"].append(r.copy())
                rec["obs"].append(o.copy())
        out = {k: np.stack(v) for k, v in rec.items()}
        out.update(grid=b.grid_center.copy(), l_cell=np.float64(b.l_cell), r_avoid=np.float64(b.r_avoid), d_sen=np.float64(b.d_sen))
        np.savez_compressed(os.path.join(HERE, f"g10_llm_n{n_a}.npz"), **out)
        print("g10_llm", n_a, "|u| max", np.abs(out["u"]).max(), "reward", out["rew"].sum((1, 2)))


def main_r3_batch():
    """Round-3 fixtures for BATCHES and the agent counts that are not powers of two: three independent reference
    episodes (own seed, own randomly drawn target shape) for each of N = 30 (the reference's default, assembly_cfg.py:153),
    100 and 200.  tests/test_gpu_parity.py::test_golden_batches runs the three episodes of one N as ONE 3-env batch with
    ragged cell sets, so the batched claims rest on the reference itself, not only on the pinned oracle."""
    gym, Wrapper = import_reference_env()
    tmp = tempfile.mkdtemp(prefix="golden_")
    pkl = os.path.join(tmp, "results.pkl")
    save_results(pkl, synthetic_shape_set())
    rng = np.random.default_rng(2263)
    for n_a, steps in ((30, 3), (100, 3), (200, 2)):
        for s, (mode, warm) in enumerate((("random", 0), ("prior", 40), ("prior", 90))):
            np.random.seed(3000 + 10 * n_a + s)
            env = make_env(gym, Wrapper, n_a, pkl)
            rec = record_episode(env, steps, mode, rng, warm)
            np.savez_compressed(os.path.join(HERE, f"g11_n{n_a}_s{s}_{mode}.npz"), **rec)
            print("g11", n_a, s, mode, "cells", rec["grid"].shape[1], "in shape", rec["in_flags"][-1].mean(), "reward", rec["rew"].sum((1, 2)))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "r3b":
        main_r3_batch()
    elif len(sys.argv) > 1 and sys.argv[1] == "r2":
        main_r2()
    elif len(sys.argv) > 1 and sys.argv[1] == "r3":
        main_r3()
    else:
        main()
