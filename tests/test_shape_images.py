"""The image -> grid-cell converter (marl_llm_amd/shape_images.py), the counterpart of the reference's cv2 pipeline
/root/reference/marl_llm/cfg/assembly_cfg.py:32-149.  No GPU.

Pins: (1) tiling / centring / scaling against a plain-loop restatement of assembly_cfg.py:56-134 written here the way the
reference iterates (tile by tile); (2) the committed cell fixture tests/golden/fig_cells.npz against a fresh conversion of
the reference's own fig/*.png when /root/reference is present (build container only); (3) the results.pkl layout round
trip through the env's loader.  The cv2 grayscale + Otsu steps cannot be compared (cv2 is not installed): parity of
those two calls is unpinned, see the module docstring."""
import os

import numpy as np
import pytest

from helpers import GOLDEN_DIR, fig_shapes


def loop_twin(binary, grid_size=36, target_height=2.2):
    """assembly_cfg.py:47-134 restated tile by tile (the vectorised product code must agree with it)."""
    black = np.argwhere(binary == 0)
    (y0, x0), (y1, x1) = black.min(axis=0), black.max(axis=0)
    b = binary[y0:y1 + 1, x0:x1 + 1]
    h, w = b.shape
    b = np.dot(np.fliplr(np.eye(h)), b)
    pts = []
    for i in range(grid_size, h - grid_size, grid_size):
        for j in range(grid_size, w - grid_size, grid_size):
            sec = b[i:i + grid_size, j:j + grid_size]
            if np.sum(sec == 0) / (grid_size * grid_size) >= 1:
                pts.append([j + grid_size / 2, i + grid_size / 2])
    pts = np.array(pts, np.float64)
    xm, ym = np.mean(pts[:, 0]), np.mean(pts[:, 1])
    pts[:, 0] -= xm; pts[:, 1] -= ym
    hs = target_height / (np.max(pts[:, 1]) - np.min(pts[:, 1]))
    ext = np.array([-0.5 - xm, w - 0.5 - xm, -0.5 - ym, h - 0.5 - ym]) * hs
    return grid_size * hs, hs * pts, b, ext


def synthetic_image(seed, h=700, w=560):
    """White page with a black blob (union of a few ellipses / bars), margins, and a grey anti-aliased rim."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.full((h, w), 255, np.uint8)
    for _ in range(4):
        cy, cx = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w
        ry, rx = rng.uniform(0.12, 0.3) * h, rng.uniform(0.12, 0.3) * w
        d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
        img[d <= 1.0] = 0
        img[(d > 1.0) & (d <= 1.03)] = np.minimum(img[(d > 1.0) & (d <= 1.03)], 120)
    return img


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_tiling_equals_the_loop_form(seed):
    from marl_llm_amd.shape_images import binarize, tile_shape
    gray = synthetic_image(seed)
    binary, t = binarize(gray)
    assert 0 < t < 255
    l1, g1, b1, e1 = tile_shape(binary)
    l2, g2, b2, e2 = loop_twin(binary)
    assert l1 == l2 and np.array_equal(g1, g2) and np.array_equal(b1, b2) and np.array_equal(e1, e2)
    assert g1.shape[0] > 20 and abs((g1[:, 1].max() - g1[:, 1].min()) - 2.2) < 1e-12
    assert abs(g1[:, 0].mean()) < 1e-12 and abs(g1[:, 1].mean()) < 1e-12


def test_otsu_separates_two_modes():
    from marl_llm_amd.shape_images import otsu_threshold
    rng = np.random.default_rng(3)
    img = np.concatenate([rng.integers(0, 30, 5000), rng.integers(200, 256, 9000)]).astype(np.uint8)
    t = otsu_threshold(img)
    assert 29 <= t < 200


def test_fig_fixture_matches_a_fresh_conversion():
    """tests/golden/fig_cells.npz == shape_images.process_folder(/root/reference/fig) (build container only)."""
    if not os.path.isdir("/root/reference/fig"):
        pytest.skip("reference figures not present (GPU box)")
    from marl_llm_amd.shape_images import process_folder
    fresh, fix = process_folder("/root/reference/fig"), fig_shapes()
    assert len(fresh["l_cell"]) == len(fix["l_cell"]) == 7
    for k in range(7):
        assert fresh["l_cell"][k] == fix["l_cell"][k] and np.array_equal(fresh["grid_coords"][k], fix["grid_coords"][k])
        assert fresh["binary_image"][k].dtype == np.float64 and fresh["shape_bound_points"][k].shape == (4,)


def test_fig_shapes_have_the_surveyed_sizes():
    """n_g 487-536 and l_cell 0.058-0.071 for the seven shipped shapes; cells form a lattice in raster
    order (rows ascending, columns ascending inside a row) -- what the env's cell index order and the lattice walk rely on."""
    fix = fig_shapes()
    n_g = [g.shape[0] for g in fix["grid_coords"]]
    assert min(n_g) == 487 and max(n_g) == 536
    assert 0.0578 < min(fix["l_cell"]) < 0.0580 and 0.0709 < max(fix["l_cell"]) < 0.0711
    for l, g in zip(fix["l_cell"], fix["grid_coords"]):
        a = (g[:, 0] - g[:, 0].min()) / l; b = (g[:, 1] - g[:, 1].min()) / l
        assert np.abs(a - np.round(a)).max() < 1e-9 and np.abs(b - np.round(b)).max() < 1e-9
        key = np.round(b) * 1000 + np.round(a)
        assert (np.diff(key) > 0).all()


def test_results_pkl_layout_round_trip(tmp_path):
    """write_results -> the file the reference's loaders open (assembly.py:113-120, eval_assembly.py:108-116)."""
    from PIL import Image
    from marl_llm_amd.shape_images import write_results
    from marl_llm_amd.shapes import load_results, r_avoid_for
    for k, seed in ((1, 0), (3, 1), (10, 2)):                 # names sort by their integer, not as strings
        Image.fromarray(synthetic_image(seed)).save(tmp_path / f"{k}.png")
    path, res = write_results(str(tmp_path))
    assert os.path.basename(path) == "results.pkl"
    back = load_results(path)
    assert set(back) == {"l_cell", "grid_coords", "binary_image", "shape_bound_points"} and len(back["l_cell"]) == 3
    from marl_llm_amd.shape_images import binarize, tile_shape
    want = tile_shape(binarize(synthetic_image(2))[0])        # third in integer order is 10.png
    assert back["l_cell"][2] == want[0] and np.array_equal(back["grid_coords"][2], want[1])
    assert r_avoid_for(30, back) > 0
    # the host env accepts the dict as args.results_file (no GPU touched before reset)
    from marl_llm_amd.env import AssemblySwarmEnv, AssemblySwarmWrapper, make_args
    env = AssemblySwarmWrapper(AssemblySwarmEnv(), make_args(n_a=8, results_file=path))
    assert env.observation_space.shape == (192, 8) and env.env.num_train_shape == 3
