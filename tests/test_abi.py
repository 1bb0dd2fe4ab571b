"""CPU-side checks of the drop-in boundary: libswarmenv.so loads without a GPU and exports every symbol
include/swarm_env.h declares; without a device the compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "swarm_env.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(?:int|void|double|const char \*)\s*\*?\s*([A-Za-z_][A-Za-z0-9_]*)\s*\(", src)
    return sorted(set(n for n in names if n.startswith("swarm_") or n.startswith("_") or n == "calculateActionPrior"))


@pytest.fixture(scope="module")
def lib():
    from marl_llm_amd.build import build_lib
    from marl_llm_amd import _lib
    build_lib()
    return _lib.load()


def test_header_symbols_exported(lib):
    names = declared_functions()
    from marl_llm_amd._lib import BATCHED_SYMBOLS, LEGACY_SYMBOLS
    assert set(BATCHED_SYMBOLS) | set(LEGACY_SYMBOLS) == set(names), names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/swarm_env.h but not exported"
    from marl_llm_amd._lib import ABI_VERSION
    assert lib.swarm_abi_version() == ABI_VERSION


def test_policy_header_symbols_exported(lib):
    src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "swarm_policy.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(swarm_policy_[a-z0-9_]+)\s*\(", src)))
    from marl_llm_amd._lib import POLICY_SYMBOLS
    assert set(names) == set(POLICY_SYMBOLS), names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/swarm_policy.h but not exported"


def test_policy_create_fails_loudly_without_a_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a HIP device")
    w = (ctypes.c_float * (192 * 192))()
    h = ctypes.c_void_p()
    rc = lib.swarm_policy_create(*[ctypes.cast(w, ctypes.c_void_p)] * 8, 192, 180, 2, -1, ctypes.byref(h))
    assert rc != 0 and b"no HIP device" in lib.swarm_policy_last_error()


def test_config_struct_matches_header(lib):
    from marl_llm_amd._lib import SwarmConfig
    cfg = SwarmConfig()
    lib.swarm_default_config(ctypes.byref(cfg))
    assert ctypes.sizeof(SwarmConfig) == 12 * 4 + 12 * 8 + 4 * 8 + 2 * 4      # + prior_gain[3], llm_repulsion, llm_action, pad
    assert (cfg.topo_nei_max, cfg.num_obs_grid_max, cfg.num_occupied_grid_max) == (6, 80, 200)
    assert (cfg.d_sen, cfg.size_a, cfg.k_ball, cfg.k_wall, cfg.c_wall, cfg.vel_max, cfg.dt) == (0.4, 0.035, 30, 100, 5, 0.8, 0.1)
    assert list(cfg.boundary) == [-2.4, 2.4, 2.4, -2.4]
    assert list(cfg.prior_gain) == [2.0, 3.0, 2.0] and cfg.llm_repulsion == 1.0 and cfg.llm_action == 0


def test_invalid_configs_rejected_with_message(lib):
    from marl_llm_amd._lib import SwarmConfig
    cfg = SwarmConfig()
    lib.swarm_default_config(ctypes.byref(cfg))
    cfg.n_agents = 1000
    h = ctypes.c_void_p()
    rc = lib.swarm_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == 1 and not h.value
    assert b"n_agents" in lib.swarm_last_error(None)


def test_no_cpu_fallback_without_device(lib):
    """On a box without a GPU the product path must fail loudly, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from marl_llm_amd._lib import SwarmConfig, SwarmError
    from marl_llm_amd.batched import SwarmBatch
    cfg = SwarmConfig()
    lib.swarm_default_config(ctypes.byref(cfg))
    h = ctypes.c_void_p()
    rc = lib.swarm_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == 2 and not h.value and b"no CPU path" in lib.swarm_last_error(None)
    with pytest.raises(SwarmError):
        SwarmBatch(n_env=1, n_agents=8, n_cells_max=16, r_avoid=0.1)
    # legacy symbol: prints an error and poisons its output instead of computing on the CPU
    p = np.zeros((2, 4)); r = np.full(4, 0.035); d = np.ones((4, 4)); c = np.zeros((4, 4), bool)
    b = np.array([-2.4, 2.4, 2.4, -2.4])
    dp_ = ctypes.POINTER(ctypes.c_double)
    lib._get_dist_b2w(p.ctypes.data_as(dp_), r.ctypes.data_as(dp_), d.ctypes.data_as(dp_),
                      c.ctypes.data_as(ctypes.POINTER(ctypes.c_bool)), ctypes.c_int(2), ctypes.c_int(4), b.ctypes.data_as(dp_))
    assert np.isnan(d).all()


def test_product_does_not_import_oracle():
    """Nothing under marl_llm_amd/ may import, link or load the oracle."""
    pkg = os.path.join(ROOT, "marl_llm_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".hip", ".h", ".cpp")):
                continue
            text = open(os.path.join(dirpath, f)).read()
            assert "liboracle" not in text and "oracle_py" not in text and "assembly_oracle" not in text, f
            for line in text.splitlines():
                assert not re.match(r"\s*(from|import)\s+oracle\b", line), (f, line)


def test_legacy_call_without_a_device_sets_the_status(lib):
    """No GPU here: the legacy symbols poison their outputs and report through swarm_legacy_status()."""
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a HIP device")
    p = np.zeros((2, 4)); d = np.zeros((4, 4)); c = np.zeros((4, 4), np.uint8)
    bnd = np.array([-2.4, 2.4, 2.4, -2.4])
    dp_ = ctypes.POINTER(ctypes.c_double)
    lib._get_dist_b2w(p.ctypes.data_as(dp_), np.full(4, 0.035).ctypes.data_as(dp_), d.ctypes.data_as(dp_),
                      c.ctypes.data_as(ctypes.POINTER(ctypes.c_bool)), ctypes.c_int(2), ctypes.c_int(4), bnd.ctypes.data_as(dp_))
    assert np.isnan(d).all() and lib.swarm_legacy_status() == 1 and b"_get_dist_b2w" in lib.swarm_legacy_last_error()
