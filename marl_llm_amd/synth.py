"""Synthetic batched inputs of the BASELINE configs' shapes (host side, numpy).

Mirrors the distribution of AssemblySwarmEnv.reset() (/root/reference/cus_gym/gym/envs/customized_envs/
assembly.py:156-219) per environment -- random shape, rotation U(-pi,pi), offset U(-1.4,1.4)^2, agents either
uniform over the arena or in a 2x2 box around a random centre (50/50), velocities U(-0.5,0.5) -- but with a
counter-based per-env generator (seed, env_id) instead of the global numpy RNG, so any env range can be
generated on any rank independently (multi-GPU sharding needs no communication).
`assembled_fraction` > 0 places that fraction of the envs' agents on the target shape (cell centre + N(0,0.05)),
which is what the swarm looks like after ~100 prior-policy steps; it exercises the occupied-cell filter.
"""
import numpy as np

HALF = 2.4


def synthetic_batch(n_env, n_agents, shapes, seed=226, assembled_fraction=0.0, env_offset=0, n_cells_max=None):
    n_shapes = len(shapes["l_cell"])
    grids = [np.asarray(g, dtype=np.float64).T for g in shapes["grid_coords"]]     # (2, n_g), assembly.py:164
    ng_max = max(g.shape[1] for g in grids) if n_cells_max is None else int(n_cells_max)
    cells = np.zeros((n_env, 2, ng_max))
    n_g = np.zeros(n_env, np.int32)
    l_cell = np.zeros(n_env)
    p = np.zeros((n_env, 2, n_agents))
    dp = np.zeros((n_env, 2, n_agents))
    for k in range(n_env):
        e = env_offset + k
        rng = np.random.default_rng([seed, e])                                     # counter-based: (seed, env id)
        s = int(rng.integers(0, n_shapes))
        th = np.pi * rng.uniform(-1, 1)                                            # assembly.py:175-178
        rot = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
        g = rot @ grids[s] + rng.uniform(-HALF + 1, HALF - 1, (2, 1))              # assembly.py:184-187
        n_g[k] = g.shape[1]; l_cell[k] = shapes["l_cell"][s]
        cells[k, :, : g.shape[1]] = g
        if rng.uniform() < assembled_fraction:
            p[k] = g[:, rng.integers(0, g.shape[1], n_agents)] + rng.normal(0, 0.05, (2, n_agents))
        elif rng.uniform(-1, 1) > 0:                                               # assembly.py:202-208
            p[k] = rng.uniform(-HALF, HALF, (2, n_agents))
        else:
            p[k] = rng.uniform(-1, 1, (2, n_agents)) + rng.uniform(-HALF + 1, HALF - 1, (2, 1))
        dp[k] = rng.uniform(-0.5, 0.5, (2, n_agents))                              # assembly.py:215
    return dict(cells=cells, n_g=n_g, l_cell=l_cell, p=p, dp=dp)
