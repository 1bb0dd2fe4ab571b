"""Shared test helpers: seeded synthetic states in the reference's layouts, golden-fixture loading."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(pattern=None):
    """Recorded reference episodes: g2_* (synthetic shape set), g6_* (the reference's own fig/*.png shapes) and g11_* (three
    episodes each of N = 30 / 100 / 200, also run as 3-env batches)."""
    pats = [pattern] if pattern else ["g2_*.npz", "g6_*.npz", "g11_*.npz"]
    return sorted(f for p in pats for f in glob.glob(os.path.join(GOLDEN_DIR, p)))


def fig_shapes():
    """The reference's seven target shapes (fig/*.png) as tiled by marl_llm_amd.shape_images; results.pkl layout."""
    from marl_llm_amd.shape_images import unpack_cells_npz
    return unpack_cells_npz(os.path.join(GOLDEN_DIR, "fig_cells.npz"))


def load_golden(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def make_case(rng, shapes, n_a, cluster, shape=None):
    """One env: rotated/offset target shape + agents either scattered over the arena or clustered on the
    shape (the latter exercises in-shape flags, the occupied-cell filter, collisions)."""
    s = int(rng.integers(0, len(shapes["l_cell"]))) if shape is None else shape
    g = shapes["grid_coords"][s].T.copy()
    l_cell = float(shapes["l_cell"][s])
    th = rng.uniform(-np.pi, np.pi)
    rot = np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
    g = np.ascontiguousarray(rot @ g + rng.uniform(-1.4, 1.4, (2, 1)))
    if cluster:
        p = g[:, rng.integers(0, g.shape[1], n_a)] + rng.normal(0, 0.05, (2, n_a))
    else:
        p = rng.uniform(-2.4, 2.4, (2, n_a))
    dp = rng.uniform(-0.5, 0.5, (2, n_a))
    return np.ascontiguousarray(p), np.ascontiguousarray(dp), g, l_cell
