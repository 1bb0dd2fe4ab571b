"""Target-shape images -> grid-cell centres: the shape pipeline in front of the env step.

The reference builds its ``results.pkl`` at import time of ``marl_llm/cfg/assembly_cfg.py`` with OpenCV
(/root/reference/marl_llm/cfg/assembly_cfg.py:32-134,139-149): grayscale read, Otsu binarisation, crop to the bounding box of
the black pixels, vertical flip, 36-pixel tiling that keeps the tiles lying entirely inside the black region, centring of
the tile centres on their mean, scaling of their height to 2.2 m, and four parallel lists ``l_cell / grid_coords /
binary_image / shape_bound_points`` pickled (:24-29,131-134).  This module restates that pipeline on PIL + numpy (cv2 is
not a dependency here), vectorised instead of looping over tiles, and writes / returns the same layout, which is what
``AssemblySwarmEnv.__reinit__`` (assembly.py:113-120) and ``eval_assembly.py:108-116`` load.

Parity note: tiling, centring and scaling are exact restatements (checked against a plain-loop twin in
tests/test_shape_images.py).  The two cv2 calls are not available to compare with: grayscale conversion uses PIL's ITU-R
601-2 luma transform (cv2 uses the same weights with its own fixed-point rounding) and the threshold is Otsu's method on
the 256-bin histogram implemented below -- "parity unpinned" for those two steps.  The reference's shapes are black
silhouettes on white, so any threshold between the two modes yields the same binary image up to anti-aliased edge pixels.
"""
import glob
import os
import pickle

import numpy as np

GRID_SIZE = 36          # assembly_cfg.py:58
TARGET_HEIGHT = 2.2     # assembly_cfg.py:96


def otsu_threshold(gray):
    """Otsu's threshold of a uint8 image: the t maximising the between-class variance of {<= t} and {> t}."""
    hist = np.bincount(np.asarray(gray, dtype=np.uint8).ravel(), minlength=256).astype(np.float64)
    total = hist.sum()
    w0 = np.cumsum(hist)                                   # pixels with value <= t
    m0 = np.cumsum(hist * np.arange(256))
    w1 = total - w0
    with np.errstate(divide="ignore", invalid="ignore"):
        mu0 = m0 / w0
        mu1 = (m0[-1] - m0) / w1
        between = w0 * w1 * (mu0 - mu1) ** 2
    between[~np.isfinite(between)] = -1.0
    return int(np.argmax(between))


def binarize(gray, threshold=None):
    """cv2.threshold(gray, t, 255, THRESH_BINARY [+ THRESH_OTSU]) (assembly_cfg.py:45): 255 where gray > t, else 0."""
    t = otsu_threshold(gray) if threshold is None else int(threshold)
    return np.where(np.asarray(gray) > t, 255, 0).astype(np.uint8), t


def tile_shape(binary_image, grid_size=GRID_SIZE, target_height=TARGET_HEIGHT):
    """Everything of process_image() after the binarisation (assembly_cfg.py:47-134), from a 0/255 image.
    Returns (l_cell, grid_coords (n_g, 2) float64, cropped + flipped image (float64, like the reference's np.dot product),
    shape_bound_points (4,))."""
    b = np.asarray(binary_image)
    ys, xs = np.nonzero(b == 0)
    if ys.size == 0:
        raise ValueError("image has no black pixel")
    b = b[ys.min():ys.max() + 1, xs.min():xs.max() + 1]                 # :48-51
    height, width = b.shape
    b = b[::-1].astype(np.float64)                                        # :54 fliplr(eye) @ image = rows reversed
    # tiles start at (i, j) = grid_size, 2 grid_size, ... < height - grid_size (:61-62); a tile that sticks out of the
    # image is shorter than grid_size^2 pixels and can never be "entirely black" (:72-76)
    i0 = np.arange(grid_size, height - grid_size, grid_size)
    j0 = np.arange(grid_size, width - grid_size, grid_size)
    i0 = i0[i0 + grid_size <= height]
    j0 = j0[j0 + grid_size <= width]
    if i0.size == 0 or j0.size == 0:
        raise ValueError("image too small for %d-pixel tiles" % grid_size)
    black = (b == 0)
    # count of black pixels per tile through a summed-area table
    sat = np.zeros((height + 1, width + 1), np.int64)
    sat[1:, 1:] = np.cumsum(np.cumsum(black, axis=0, dtype=np.int64), axis=1)
    I, J = np.meshgrid(i0, j0, indexing="ij")
    cnt = sat[I + grid_size, J + grid_size] - sat[I, J + grid_size] - sat[I + grid_size, J] + sat[I, J]
    full = cnt == grid_size * grid_size                                    # black_pixel_ratio >= 1 (:76)
    if not full.any():
        raise ValueError("no %d-pixel tile lies entirely inside the shape" % grid_size)
    # raster order: i (rows) outer, j inner = the reference's loop order, which is the cell index order of the env
    coords = np.stack([J[full] + grid_size / 2, I[full] + grid_size / 2], axis=1).astype(np.float64)   # (x, y) :66-67
    x_mean, y_mean = np.mean(coords[:, 0]), np.mean(coords[:, 1])          # :84-87
    coords[:, 0] -= x_mean
    coords[:, 1] -= y_mean
    h_scale = target_height / (np.max(coords[:, 1]) - np.min(coords[:, 1]))   # :96-98
    grid_coords = h_scale * coords
    # imshow(origin='lower') extent of an (height, width) image is (-0.5, width-0.5, -0.5, height-0.5) (:105-111)
    ext = np.array([-0.5 - x_mean, width - 0.5 - x_mean, -0.5 - y_mean, height - 0.5 - y_mean])
    return float(grid_size * h_scale), grid_coords, b, ext * h_scale       # :128-134


def process_image(image_path, threshold=None):
    """One image file -> (l_cell, grid_coords, binary_image, shape_bound_points); assembly_cfg.py:32-134."""
    from PIL import Image
    with Image.open(image_path) as im:
        gray = np.asarray(im.convert("L"))                                 # cv2.imread(..., IMREAD_GRAYSCALE) :44
    binary, _ = binarize(gray, threshold)
    return tile_shape(binary)


def image_paths(folder):
    """The reference's file order: *.png sorted by the integer in the file name (assembly_cfg.py:139-140)."""
    return sorted(glob.glob(os.path.join(folder, "*.png")), key=lambda x: int(os.path.basename(x).split(".")[0]))


def process_folder(folder, threshold=None):
    """All images of a folder -> dict in the results.pkl layout (assembly_cfg.py:24-29,143-149)."""
    res = {"l_cell": [], "grid_coords": [], "binary_image": [], "shape_bound_points": []}
    for path in image_paths(folder):
        l_cell, coords, img, bounds = process_image(path, threshold)
        res["l_cell"].append(l_cell)
        res["grid_coords"].append(coords)
        res["binary_image"].append(img)
        res["shape_bound_points"].append(bounds)
    if not res["l_cell"]:
        raise FileNotFoundError("no <integer>.png image in %r" % (folder,))
    return res


def write_results(folder, out_path=None, threshold=None):
    """process_folder + pickle.dump to <folder>/results.pkl (assembly_cfg.py:146-149).  Returns (path, results)."""
    res = process_folder(folder, threshold)
    out_path = out_path or os.path.join(folder, "results.pkl")
    with open(out_path, "wb") as f:
        pickle.dump(res, f)
    return out_path, res


def pack_cells_npz(results, path):
    """Compact, pickle-free form of the step-relevant half of a results dict (l_cell + grid_coords): test fixtures."""
    n = np.array([np.asarray(g).shape[0] for g in results["grid_coords"]], np.int64)
    np.savez_compressed(path, l_cell=np.asarray(results["l_cell"], np.float64), n_g=n,
                        coords=np.concatenate([np.asarray(g, np.float64) for g in results["grid_coords"]], axis=0))


def unpack_cells_npz(path):
    with np.load(path, allow_pickle=False) as z:
        l_cell, n, coords = z["l_cell"], z["n_g"], z["coords"]
    off = np.concatenate([[0], np.cumsum(n)])
    grids = [np.ascontiguousarray(coords[off[k]:off[k + 1]]) for k in range(len(n))]
    return {"l_cell": [float(v) for v in l_cell], "grid_coords": grids,
            "binary_image": [np.zeros((2, 2)) for _ in grids],
            "shape_bound_points": [np.array([g[:, 0].min(), g[:, 0].max(), g[:, 1].min(), g[:, 1].max()]) for g in grids]}
