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

Usage:  MPLBACKEND=Agg python tests/golden/make_golden.py
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


if __name__ == "__main__":
    main()
