"""Target-shape sets for the batched assembly env.

The reference turns silhouette images into grid-cell centres with OpenCV at import time
(/root/reference/marl_llm/cfg/assembly_cfg.py:32-134: 36-px tiling of the black region, centring on the
mean, scaling the cell-centre height to 2.2 m) and stores them in a pickle with four parallel lists
``l_cell, grid_coords, binary_image, shape_bound_points`` (assembly_cfg.py:24-29,131-134,147-149) that
``AssemblySwarmEnv.__reinit__`` loads (assembly.py:113-120).  The images' ``results.pkl`` is not shipped,
and cv2 is not available here, so this module provides

* :func:`synthetic_shape_set` -- seven deterministic geometric silhouettes tiled the same way
  (480-540 cells, ``l_cell`` 0.058-0.071, cell-centre height 2.2) used by bench / tests, and
* :func:`load_results` / :func:`save_results` for files in the reference's ``results.pkl`` layout.

Only ``l_cell`` and ``grid_coords`` influence the env step (assembly.py:116-124,163-164).
"""
import pickle

import numpy as np

TARGET_HEIGHT = 2.2  # assembly_cfg.py:96


def _inside(name, x, y):
    """Silhouette predicates on the unit square [-1,1]^2 (y up)."""
    if name == "ellipse":
        return (x / 0.5) ** 2 + y * y <= 1.0
    if name == "rect":
        return (np.abs(x) <= 0.42) & (np.abs(y) <= 1.0)
    if name == "L":
        return (((x >= -0.6) & (x <= -0.1) & (np.abs(y) <= 1.0))
                | ((y >= -1.0) & (y <= -0.5) & (x >= -0.6) & (x <= 0.9)))
    if name == "T":
        return (((y >= 0.55) & (y <= 1.0) & (np.abs(x) <= 0.9))
                | ((np.abs(x) <= 0.25) & (y >= -1.0) & (y <= 1.0)))
    if name == "plus":
        return ((np.abs(x) <= 0.23) & (np.abs(y) <= 1.0)) | ((np.abs(y) <= 0.23) & (np.abs(x) <= 1.0))
    if name == "ring":
        r2 = x * x + y * y
        return (r2 <= 1.0) & (r2 >= 0.5)
    if name == "triangle":
        return (y >= -1.0) & (y <= 1.0) & (np.abs(x) <= (1.0 - y) * 0.41)
    raise ValueError(name)


SHAPE_NAMES = ("ellipse", "rect", "L", "T", "plus", "ring", "triangle")


def make_shape(name, n_lo=480, n_hi=540):
    """Tile silhouette `name`; returns (l_cell, grid_coords (n_g,2) float64) with n_lo <= n_g <= n_hi."""
    best, best_err = None, None
    for k in range(12, 120):                      # cells across the unit square's height
        h = 2.0 / k
        # cell centres of a raster scan (row-major: y outer, x inner) like assembly_cfg.py:61-77
        c = -1.0 + h * (np.arange(k) + 0.5)
        yy, xx = np.meshgrid(c, c, indexing="ij")
        # a cell belongs to the shape iff its four corners and centre are inside ("entirely black")
        ok = _inside(name, xx, yy)
        for sx in (-0.5, 0.5):
            for sy in (-0.5, 0.5):
                ok &= _inside(name, xx + sx * h, yy + sy * h)
        n = int(ok.sum())
        err = abs(n - (n_lo + n_hi) // 2)
        if n_lo <= n <= n_hi and (best is None or err < best_err):
            best, best_err = (h, xx[ok], yy[ok]), err
        if n > 2 * n_hi:
            break
    if best is None:
        raise RuntimeError(f"no tiling of {name} with {n_lo}..{n_hi} cells")
    h, xs, ys = best
    coords = np.stack([xs, ys], axis=1).astype(np.float64)
    coords[:, 0] -= coords[:, 0].mean()          # assembly_cfg.py:84-87
    coords[:, 1] -= coords[:, 1].mean()
    scale = TARGET_HEIGHT / (coords[:, 1].max() - coords[:, 1].min())   # assembly_cfg.py:96-99
    return float(h * scale), coords * scale


def synthetic_shape_set(names=SHAPE_NAMES):
    """dict in the reference's results.pkl layout."""
    res = {"l_cell": [], "grid_coords": [], "binary_image": [], "shape_bound_points": []}
    for nm in names:
        l_cell, coords = make_shape(nm)
        res["l_cell"].append(l_cell)
        res["grid_coords"].append(coords)
        res["binary_image"].append(np.zeros((2, 2)))
        res["shape_bound_points"].append(np.array([coords[:, 0].min() - l_cell, coords[:, 0].max() + l_cell,
                                                   coords[:, 1].min() - l_cell, coords[:, 1].max() + l_cell]))
    return res


def save_results(path, results):
    with open(path, "wb") as f:
        pickle.dump(results, f)


def load_results(path):
    """Load a results.pkl written by :func:`save_results` or by the reference's cfg module.
    (Unpickling executes code from the file: only load files you produced.)"""
    with open(path, "rb") as f:
        res = pickle.load(f)
    for k in ("l_cell", "grid_coords"):
        if k not in res:
            raise KeyError(f"{path}: missing key {k!r} (expected the reference's results.pkl layout)")
    return res


def r_avoid_for(n_a, results):
    """Collision-avoidance radius, assembly.py:123-124."""
    n_gs = [np.asarray(g).shape[0] for g in results["grid_coords"]]
    return round(float(np.sqrt(4 * np.min(n_gs) / (n_a * np.pi)) * np.min(results["l_cell"])), 2)
