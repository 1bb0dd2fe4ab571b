"""The reference's five extern "C" symbols served by libswarmenv.so (GPU-backed), called exactly the way
assembly.py calls them (host numpy buffers, ctypes), compared bit for bit with the oracle -- BASELINE config 0
("8 agents x 1 env, plumbing") plus a few larger sizes."""
import ctypes

import numpy as np
import pytest

from helpers import make_case

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def legacy():
    """RefLib-style caller bound to OUR library instead of libAssemblyEnv.so."""
    from marl_llm_amd import _lib
    from oracle.oracle_py import RefLib
    r = RefLib.__new__(RefLib)
    r.lib = _lib.load()
    return r


@pytest.mark.parametrize("n_a,cluster,periodic,with_self", [(8, 1, False, True), (8, 0, True, True), (8, 1, False, False),
                                                            (30, 1, False, True), (64, 1, True, True), (200, 1, False, True)])
def test_legacy_symbols_match_oracle(legacy, oracle, shapes, n_a, cluster, periodic, with_self):
    from marl_llm_amd.shapes import r_avoid_for
    from oracle.oracle_py import ref_step
    rng = np.random.default_rng(99 + n_a)
    ra = r_avoid_for(n_a, shapes)
    for rep in range(2):
        p, dp, g, l_cell = make_case(rng, shapes, n_a, cluster)
        a = oracle.get_observation(p, dp, g, l_cell, ra, is_periodic=periodic, with_self=with_self)
        b = legacy.get_observation(p, dp, g, l_cell, ra, is_periodic=periodic, with_self=with_self)
        for k in a:
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(oracle.get_reward(p, g, a["neighbor_index"], a["in_flags"], a["sensed_index"], ra, is_periodic=periodic),
                              legacy.get_reward(p, g, a["neighbor_index"], a["in_flags"], a["sensed_index"], ra, is_periodic=periodic,
                                                occupied_index=a["occupied_index"]))
        assert np.array_equal(oracle.action_prior(p, dp, g, a["neighbor_index"], l_cell, ra),
                              legacy.action_prior(p, dp, g, a["neighbor_index"], l_cell, ra))
        dc, de, co = oracle.dist_b2b(p, is_periodic=periodic)
        assert np.array_equal(oracle.sf_b2b_all(p, de, co, dc, is_periodic=periodic),
                              legacy.sf_b2b_all(p, de, co, dc, is_periodic=periodic))
        wa, wb = oracle.dist_b2w(p), legacy.dist_b2w(p)
        assert np.array_equal(wa[0], wb[0]) and np.array_equal(wa[1], wb[1])
        # a whole reference-style step (numpy glue + five native calls) served by our library
        act = rng.uniform(-1, 1, (2, n_a)).astype(np.float32)
        s1 = oracle.step(p, dp, act, g, a["neighbor_index"], l_cell, ra, is_boundary=not periodic, with_self=with_self)
        s2 = ref_step(legacy, p, dp, act, g, a["neighbor_index"], l_cell, ra, is_boundary=not periodic, with_self=with_self)
        for k in s1:
            assert np.array_equal(s1[k], s2[k]), k


def test_legacy_failure_is_reported(legacy, shapes):
    """A configuration the kernels do not support (n_a > 256) must not pass silently: NaN outputs, a message on stderr
    and a non-zero swarm_legacy_status() with the reason."""
    from marl_llm_amd.shapes import r_avoid_for
    rng = np.random.default_rng(0)
    p, dp, g, l_cell = make_case(rng, shapes, 300, 0)
    b = legacy.get_observation(p, dp, g, l_cell, r_avoid_for(300, shapes))
    assert np.isnan(b["obs"]).all()
    assert legacy.lib.swarm_legacy_status() == 1 and b"n_agents must be in [1, 256]" in legacy.lib.swarm_legacy_last_error()
    p, dp, g, l_cell = make_case(rng, shapes, 8, 0)
    legacy.get_observation(p, dp, g, l_cell, r_avoid_for(8, shapes))
    assert legacy.lib.swarm_legacy_status() == 0
