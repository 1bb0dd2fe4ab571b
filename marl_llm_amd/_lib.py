"""ctypes binding of libswarmenv.so (include/swarm_env.h).  No fallback: if the library is missing or
no HIP device is present, every use fails loudly."""
import ctypes
import os

from .build import LIB

F32, F64, BF16 = 0, 1, 2


class SwarmConfig(ctypes.Structure):
    _fields_ = [("n_env", ctypes.c_int32), ("n_agents", ctypes.c_int32), ("n_cells_max", ctypes.c_int32),
                ("topo_nei_max", ctypes.c_int32), ("num_obs_grid_max", ctypes.c_int32),
                ("num_occupied_grid_max", ctypes.c_int32), ("is_boundary", ctypes.c_int32),
                ("with_self_state", ctypes.c_int32), ("with_prior", ctypes.c_int32), ("obs_dtype", ctypes.c_int32),
                ("device", ctypes.c_int32), ("debug_flags", ctypes.c_int32),
                ("d_sen", ctypes.c_double), ("r_avoid", ctypes.c_double), ("size_a", ctypes.c_double),
                ("k_ball", ctypes.c_double), ("k_wall", ctypes.c_double), ("c_wall", ctypes.c_double),
                ("vel_max", ctypes.c_double), ("dt", ctypes.c_double), ("boundary", ctypes.c_double * 4),
                ("prior_gain", ctypes.c_double * 3), ("llm_repulsion", ctypes.c_double),
                ("llm_action", ctypes.c_int32), ("reserved_", ctypes.c_int32)]


class HostOut(ctypes.Structure):                # include/swarm_env.h swarm_host_out_t
    _fields_ = [("obs", ctypes.POINTER(ctypes.c_double)), ("a_prior", ctypes.POINTER(ctypes.c_double)),
                ("reward", ctypes.POINTER(ctypes.c_double)), ("done", ctypes.POINTER(ctypes.c_uint8))]


class SwarmError(RuntimeError):
    pass


_LIB = None

ABI_VERSION = 4            # include/swarm_env.h SWARM_ABI_VERSION
BATCHED_SYMBOLS = ("swarm_abi_version", "swarm_default_config", "swarm_create", "swarm_destroy", "swarm_last_error",
                   "swarm_set_stream", "swarm_synchronize", "swarm_obs_dim", "swarm_set_cells", "swarm_set_state",
                   "swarm_get_state", "swarm_observe", "swarm_step", "swarm_get_indices",
                   "swarm_step_algorithmic_bytes", "swarm_timer_start", "swarm_timer_stop", "swarm_lattice_envs", "swarm_set_shapes", "swarm_reset", "swarm_get_cells", "swarm_get_shape_index", "swarm_metrics", "swarm_rule_action",
                   "swarm_host_outputs", "swarm_observe_host", "swarm_step_host", "swarm_get_llm_action")
POLICY_SYMBOLS = ("swarm_policy_create", "swarm_policy_destroy", "swarm_policy_forward", "swarm_policy_forward_bf16",
                  "swarm_policy_forward_explore", "swarm_policy_set_precision", "swarm_policy_last_error")   # include/swarm_policy.h
LEGACY_SYMBOLS = ("_get_observation", "_get_reward", "_sf_b2b_all", "_get_dist_b2w", "calculateActionPrior",
                  "swarm_legacy_status", "swarm_legacy_last_error")


def load():
    """Load libswarmenv.so (built in-tree by marl_llm_amd.build / __graft_entry__.build())."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("SWARM_LIB", LIB)      # SWARM_LIB: diagnostic builds only (tools/)
    if not os.path.exists(path):
        raise SwarmError(f"{path} not found: build it with `python -m marl_llm_amd.build` "
                         "(there is no CPU fallback for the env step)")
    lib = ctypes.CDLL(path)
    vp, i32, dbl = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.swarm_abi_version.restype = i32
    lib.swarm_default_config.argtypes = [ctypes.POINTER(SwarmConfig)]; lib.swarm_default_config.restype = None
    lib.swarm_create.argtypes = [ctypes.POINTER(SwarmConfig), ctypes.POINTER(vp)]; lib.swarm_create.restype = i32
    lib.swarm_destroy.argtypes = [vp]; lib.swarm_destroy.restype = i32
    lib.swarm_last_error.argtypes = [vp]; lib.swarm_last_error.restype = ctypes.c_char_p
    lib.swarm_set_stream.argtypes = [vp, vp]; lib.swarm_set_stream.restype = i32
    lib.swarm_synchronize.argtypes = [vp]; lib.swarm_synchronize.restype = i32
    lib.swarm_obs_dim.argtypes = [vp]; lib.swarm_obs_dim.restype = i32
    lib.swarm_set_cells.argtypes = [vp, i32, i32, vp, vp, vp]; lib.swarm_set_cells.restype = i32
    lib.swarm_set_state.argtypes = [vp, vp, vp]; lib.swarm_set_state.restype = i32
    lib.swarm_get_state.argtypes = [vp, vp, vp]; lib.swarm_get_state.restype = i32
    lib.swarm_get_cells.argtypes = [vp, vp, vp]; lib.swarm_get_cells.restype = i32
    lib.swarm_get_shape_index.argtypes = [vp, vp]; lib.swarm_get_shape_index.restype = i32
    lib.swarm_metrics.argtypes = [vp, vp]; lib.swarm_metrics.restype = i32
    lib.swarm_legacy_status.argtypes = []; lib.swarm_legacy_status.restype = i32
    lib.swarm_legacy_last_error.argtypes = []; lib.swarm_legacy_last_error.restype = ctypes.c_char_p
    lib.swarm_rule_action.argtypes = [vp, vp]; lib.swarm_rule_action.restype = i32
    lib.swarm_policy_create.argtypes = [vp] * 8 + [i32] * 4 + [ctypes.POINTER(vp)]; lib.swarm_policy_create.restype = i32
    lib.swarm_policy_destroy.argtypes = [vp]; lib.swarm_policy_destroy.restype = None
    lib.swarm_policy_forward.argtypes = [vp, vp, ctypes.c_int64, vp, vp]; lib.swarm_policy_forward.restype = i32
    lib.swarm_policy_forward_bf16.argtypes = [vp, vp, ctypes.c_int64, vp, vp]; lib.swarm_policy_forward_bf16.restype = i32
    lib.swarm_policy_forward_explore.argtypes = [vp, vp, i32, ctypes.c_int64, vp, ctypes.c_float, ctypes.c_uint64, ctypes.c_uint64, vp]
    lib.swarm_policy_forward_explore.restype = i32
    lib.swarm_policy_set_precision.argtypes = [vp, i32]; lib.swarm_policy_set_precision.restype = i32
    lib.swarm_policy_last_error.argtypes = []; lib.swarm_policy_last_error.restype = ctypes.c_char_p
    lib.swarm_observe.argtypes = [vp, vp]; lib.swarm_observe.restype = i32
    lib.swarm_step.argtypes = [vp, vp, i32, vp, vp, vp, vp]; lib.swarm_step.restype = i32
    lib.swarm_get_indices.argtypes = [vp, vp, vp, vp, vp]; lib.swarm_get_indices.restype = i32
    lib.swarm_step_algorithmic_bytes.argtypes = [vp]; lib.swarm_step_algorithmic_bytes.restype = dbl
    lib.swarm_lattice_envs.argtypes = [vp]; lib.swarm_lattice_envs.restype = i32
    lib.swarm_set_shapes.argtypes = [vp, i32, vp, vp, vp]; lib.swarm_set_shapes.restype = i32
    lib.swarm_reset.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int64, vp]; lib.swarm_reset.restype = i32
    lib.swarm_host_outputs.argtypes = [vp, i32, ctypes.POINTER(HostOut)]; lib.swarm_host_outputs.restype = i32
    lib.swarm_observe_host.argtypes = [vp, i32]; lib.swarm_observe_host.restype = i32
    lib.swarm_step_host.argtypes = [vp, vp, i32, i32, i32]; lib.swarm_step_host.restype = i32
    lib.swarm_get_llm_action.argtypes = [vp, vp]; lib.swarm_get_llm_action.restype = i32
    lib.swarm_timer_start.argtypes = [vp]; lib.swarm_timer_start.restype = i32
    lib.swarm_timer_stop.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]; lib.swarm_timer_stop.restype = i32
    if lib.swarm_abi_version() != ABI_VERSION:
        raise SwarmError("libswarmenv.so ABI version mismatch; rebuild")
    _LIB = lib
    return lib


def check(lib, handle, rc):
    if rc != 0:
        msg = lib.swarm_last_error(handle)
        raise SwarmError(f"libswarmenv error {rc}: {msg.decode() if msg else '?'}")
