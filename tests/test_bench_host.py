"""Host-side pieces of bench.py that need no GPU: the CPU-baseline worker pool (independent processes stepping the
reference's CPU path over their own envs) and the SURVEY 8(d) byte accounting."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_worker_pool(tmp_path, shapes, oracle):
    import bench
    from marl_llm_amd.shapes import r_avoid_for
    from marl_llm_amd.synth import synthetic_batch
    E, N = 6, 16
    ra = r_avoid_for(N, shapes)
    sy = synthetic_batch(E, N, shapes, seed=1, assembled_fraction=0.5)
    nei = np.stack([oracle.get_observation(sy["p"][e], sy["dp"][e], sy["cells"][e][:, : sy["n_g"][e]],
                                           float(sy["l_cell"][e]), ra)["neighbor_index"] for e in range(E)])
    path = str(tmp_path / "state.npz")
    np.savez(path, p=sy["p"], dp=sy["dp"], nei=nei, cells=sy["cells"], n_g=sy["n_g"], l_cell=sy["l_cell"])
    pool = bench.start_cpu_workers(2)
    r = bench.run_cpu_leg(pool, path, N, ra, budget_s=1.0, steps_per_env=5)
    assert r["value"] > 0 and 2 <= r["envs"] <= E and r["kind"] in ("reference", "port")
    assert all(p.returncode == 0 for p in pool)


def test_survey_bytes_matches_section_8d():
    import bench
    # N = 64, n_g = 512 -> 885 B per agent-step
    b = bench.survey_bytes(64, np.full(4096, 512))
    assert b == 4096 * 64 * (821 + 8 * 512 / 64) and abs(b / (4096 * 64) - 885) < 1e-9
    assert bench.host_cores() >= 1
