"""SwarmBatch: thin torch-facing wrapper around one libswarmenv handle (one GPU, E environments).

Device memory, streams and tensors come from PyTorch-ROCm; all compute is in the HIP library.
Layouts are the ones documented in include/swarm_env.h.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F32, F64, SwarmConfig, SwarmError, check


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class SwarmBatch:
    def __init__(self, n_env, n_agents, n_cells_max, r_avoid, *, is_boundary=True, with_self=True, with_prior=True,
                 obs_dtype=torch.float32, device="cuda:0", d_sen=0.4, topo=6, g_max=80, occ_max=200,
                 boundary=(-2.4, 2.4, 2.4, -2.4), size_a=0.035, k_ball=30.0, k_wall=100.0, c_wall=5.0,
                 vel_max=0.8, dt=0.1, debug_flags=0, llm_action=False, prior_gain=(2.0, 3.0, 2.0), llm_repulsion=1.0):
        if not torch.cuda.is_available():
            raise SwarmError("no HIP device visible to PyTorch: the env step has no CPU fallback")
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise SwarmError("SwarmBatch needs a cuda (HIP) device")
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        cfg = SwarmConfig()
        self.lib.swarm_default_config(ctypes.byref(cfg))
        cfg.n_env, cfg.n_agents, cfg.n_cells_max = int(n_env), int(n_agents), int(n_cells_max)
        cfg.topo_nei_max, cfg.num_obs_grid_max, cfg.num_occupied_grid_max = int(topo), int(g_max), int(occ_max)
        cfg.is_boundary, cfg.with_self_state, cfg.with_prior = int(bool(is_boundary)), int(bool(with_self)), int(bool(with_prior))
        if obs_dtype not in (torch.float32, torch.float64, torch.bfloat16):
            raise SwarmError("obs_dtype must be torch.float32, torch.float64 or torch.bfloat16")
        cfg.obs_dtype = F64 if obs_dtype == torch.float64 else BF16 if obs_dtype == torch.bfloat16 else F32
        cfg.device = dev_index
        cfg.debug_flags = int(debug_flags)
        cfg.d_sen, cfg.r_avoid, cfg.size_a = float(d_sen), float(r_avoid), float(size_a)
        cfg.k_ball, cfg.k_wall, cfg.c_wall, cfg.vel_max, cfg.dt = float(k_ball), float(k_wall), float(c_wall), float(vel_max), float(dt)
        for k in range(4):
            cfg.boundary[k] = float(boundary[k])
        for k in range(3):
            cfg.prior_gain[k] = float(prior_gain[k])
        cfg.llm_repulsion, cfg.llm_action = float(llm_repulsion), int(bool(llm_action))
        self.has_llm_action = bool(llm_action)
        self._host = None
        self.cfg = cfg
        self.handle = ctypes.c_void_p()
        rc = self.lib.swarm_create(ctypes.byref(cfg), ctypes.byref(self.handle))
        if rc != 0:
            raise SwarmError(f"swarm_create failed ({rc}): {self.lib.swarm_last_error(None).decode()}")
        self.n_env, self.n_agents, self.n_cells_max = int(n_env), int(n_agents), int(n_cells_max)
        self.topo, self.g_max, self.occ_max = int(topo), int(g_max), int(occ_max)
        self.obs_dtype = obs_dtype
        self.with_prior = bool(with_prior)
        self.obs_dim = self.lib.swarm_obs_dim(self.handle)
        E, N, D = self.n_env, self.n_agents, self.obs_dim
        # ping-pong output buffers: the previous step's tensors stay valid for one more step
        self._obs = [torch.empty((E, N, D), dtype=obs_dtype, device=self.device) for _ in range(2)]
        self._rew = [torch.empty((E, N), dtype=torch.float32, device=self.device) for _ in range(2)]
        self._pri = [torch.empty((E, N, 2), dtype=obs_dtype, device=self.device) for _ in range(2)]
        self._done = torch.zeros((E, N), dtype=torch.uint8, device=self.device)
        self._flip = 0

    # -- plumbing -----------------------------------------------------------------------------------
    def _sync_stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream
        check(self.lib, self.handle, self.lib.swarm_set_stream(self.handle, ctypes.c_void_p(s)))

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            self.lib.swarm_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setup --------------------------------------------------------------------------------------
    def set_cells(self, cells, n_g, l_cell, env_begin=0):
        """cells [count, 2, n_cells_max] float64 (numpy or torch, host or device); n_g, l_cell [count] (host)."""
        n_g = np.ascontiguousarray(n_g, dtype=np.int32)
        l_cell = np.ascontiguousarray(l_cell, dtype=np.float64)
        count = int(n_g.shape[0])
        if isinstance(cells, torch.Tensor):
            c = cells.to(dtype=torch.float64).contiguous()
            if tuple(c.shape) != (count, 2, self.n_cells_max):
                raise SwarmError(f"cells must be [{count}, 2, {self.n_cells_max}]")
            cptr = _ptr(c)
        else:
            c = np.ascontiguousarray(cells, dtype=np.float64)
            if c.shape != (count, 2, self.n_cells_max):
                raise SwarmError(f"cells must be [{count}, 2, {self.n_cells_max}]")
            cptr = c.ctypes.data_as(ctypes.c_void_p)
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_set_cells(self.handle, int(env_begin), count, cptr,
                                                              n_g.ctypes.data_as(ctypes.c_void_p),
                                                              l_cell.ctypes.data_as(ctypes.c_void_p)))

    def set_shapes(self, results):
        """Upload a shape set (the reference's results.pkl layout: 'grid_coords' list of (n_g, 2), 'l_cell' list)
        for the device-side reset."""
        grids = [np.asarray(g, dtype=np.float64).T for g in results["grid_coords"]]       # (2, n_g), assembly.py:164
        S = len(grids)
        cells = np.zeros((S, 2, self.n_cells_max))
        n_g = np.zeros(S, np.int32)
        for k, g in enumerate(grids):
            if g.shape[1] > self.n_cells_max:
                raise SwarmError("shape %d has %d cells > n_cells_max=%d" % (k, g.shape[1], self.n_cells_max))
            n_g[k] = g.shape[1]; cells[k, :, : g.shape[1]] = g
        l_cell = np.ascontiguousarray(results["l_cell"], dtype=np.float64)
        check(self.lib, self.handle, self.lib.swarm_set_shapes(self.handle, S, cells.ctypes.data_as(ctypes.c_void_p),
                                                               n_g.ctypes.data_as(ctypes.c_void_p),
                                                               l_cell.ctypes.data_as(ctypes.c_void_p)))

    def reset(self, seed, episode=0, env_offset=0):
        """Batched device-side reset (swarm_reset): returns the first observation tensor."""
        self._flip ^= 1
        obs = self._obs[self._flip]
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_reset(self.handle, int(seed), int(episode), int(env_offset), _ptr(obs)))
        return obs

    def get_cells(self):
        """(cells [E,2,n_cells_max] float64, n_g [E] int32) currently on the device (numpy copies)."""
        cells = np.empty((self.n_env, 2, self.n_cells_max)); n_g = np.empty(self.n_env, np.int32)
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_get_cells(self.handle, cells.ctypes.data_as(ctypes.c_void_p),
                                                              n_g.ctypes.data_as(ctypes.c_void_p)))
        return cells, n_g

    def get_shape_index(self):
        """[E] int32 numpy: the shape index every env drew in the last device-side reset (-1: cells set by hand)."""
        out = np.empty(self.n_env, np.int32)
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_get_shape_index(self.handle, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def set_state(self, p, dp):
        """p, dp [E, 2, N] float64 (numpy or torch)."""
        keep = []
        ptrs = []
        for a in (p, dp):
            if isinstance(a, torch.Tensor):
                t = a.to(dtype=torch.float64).contiguous()
                shape = tuple(t.shape); ptrs.append(_ptr(t))
            else:
                t = np.ascontiguousarray(a, dtype=np.float64)
                shape = t.shape; ptrs.append(t.ctypes.data_as(ctypes.c_void_p))
            if shape != (self.n_env, 2, self.n_agents):
                raise SwarmError(f"state arrays must be [{self.n_env}, 2, {self.n_agents}]")
            keep.append(t)
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_set_state(self.handle, ptrs[0], ptrs[1]))

    def get_state(self):
        p = torch.empty((self.n_env, 2, self.n_agents), dtype=torch.float64, device=self.device)
        dp = torch.empty_like(p)
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_get_state(self.handle, _ptr(p), _ptr(dp)))
        return p, dp

    # -- hot path -----------------------------------------------------------------------------------
    def observe(self):
        """Recompute obs + caches from the current state (the tail of the reference's reset())."""
        self._flip ^= 1
        obs = self._obs[self._flip]
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_observe(self.handle, _ptr(obs)))
        return obs

    def _out_ptrs(self, out):
        """Caller-owned output tensors (e.g. the slots of a replay ring: the step then writes its transition where it is
        kept): dict with any of obs [E,N,D] (obs dtype), rew [E,N] f32, done [E,N] uint8, prior [E,N,2] (obs dtype)."""
        E, N, D = self.n_env, self.n_agents, self.obs_dim
        want = {"obs": ((E, N, D), self.obs_dtype), "rew": ((E, N), torch.float32), "done": ((E, N), torch.uint8),
                "prior": ((E, N, 2), self.obs_dtype)}
        res = {}
        for k, (shape, dt) in want.items():
            t = out.get(k)
            if t is None:
                continue
            if t.device != self.device or t.dtype != dt or t.numel() != int(np.prod(shape)) or not t.is_contiguous():
                raise SwarmError(f"out[{k!r}] must be a contiguous {dt} tensor of {int(np.prod(shape))} elements on {self.device}")
            res[k] = t
        return res

    def step(self, action, out=None):
        """action [E, N, 2] float32/float64 device tensor -> (obs [E,N,D], reward [E,N], done [E,N] uint8,
        a_prior [E,N,2] or None).  Asynchronous on torch's current stream.  out: optional dict of caller-owned output
        tensors (see _out_ptrs) written instead of the batch's own ping-pong buffers."""
        if action is None:
            if not self.has_llm_action:
                raise SwarmError("action=None needs a batch created with llm_action=True (agent_strategy 'llm')")
        else:
            if not isinstance(action, torch.Tensor) or action.device != self.device:
                raise SwarmError("action must be a torch tensor on the env's device")
            if tuple(action.shape) != (self.n_env, self.n_agents, 2):
                raise SwarmError(f"action must be [{self.n_env}, {self.n_agents}, 2]")
            if action.dtype not in (torch.float32, torch.float64):
                action = action.to(torch.float32)
            action = action.contiguous()
        self._flip ^= 1
        f = self._flip
        o = self._out_ptrs(out) if out else {}
        E, N, D = self.n_env, self.n_agents, self.obs_dim
        obs = o["obs"].view(E, N, D) if "obs" in o else self._obs[f]
        rew = o["rew"].view(E, N) if "rew" in o else self._rew[f]
        done = o["done"].view(E, N) if "done" in o else self._done
        pri = (o["prior"].view(E, N, 2) if "prior" in o else self._pri[f]) if self.with_prior else None
        self._sync_stream()
        check(self.lib, self.handle,
              self.lib.swarm_step(self.handle, _ptr(action), F64 if (action is None or action.dtype == torch.float64) else F32,
                                  _ptr(obs), _ptr(rew), _ptr(done), _ptr(pri)))
        return obs, rew, done, pri

    # -- the reference-shaped host outputs (numpy API) -----------------------------------------------------
    def host_views(self):
        """Numpy views of the library's two pinned output slots, in the reference's layouts: per slot a dict obs (D, E*N)
        f64, a_prior (2, E*N) f64, reward (1, E*N) f64, done (1, E*N) bool.  Created once; the arrays are rewritten in
        place by step_host / observe_host (ping-pong: a slot is reused every second call)."""
        if self._host is None:
            from ._lib import HostOut
            EN, D = self.n_env * self.n_agents, self.obs_dim
            slots = []
            for s in (0, 1):
                o = HostOut()
                check(self.lib, self.handle, self.lib.swarm_host_outputs(self.handle, s, ctypes.byref(o)))
                slots.append(dict(obs=np.ctypeslib.as_array(o.obs, shape=(D, EN)),
                                  a_prior=np.ctypeslib.as_array(o.a_prior, shape=(2, EN)),
                                  reward=np.ctypeslib.as_array(o.reward, shape=(1, EN)),
                                  done=np.ctypeslib.as_array(o.done, shape=(1, EN)).view(np.bool_)))
            self._host = slots
            self._hslot = 0
        return self._host

    def observe_host(self):
        """swarm_observe + obs in the reference's (D, E*N) float64 host layout (a view of a pinned slot)."""
        h = self.host_views()
        self._hslot ^= 1
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_observe_host(self.handle, self._hslot))
        return h[self._hslot]["obs"]

    def step_host(self, action):
        """One step through the host boundary.  action: numpy (2, E*N) float32/float64 in the reference's layout, a device
        tensor [E, N, 2], or None (llm_action batches).  Returns the slot dict of host_views()."""
        h = self.host_views()
        self._hslot ^= 1
        self._sync_stream()
        if action is None:
            rc = self.lib.swarm_step_host(self.handle, None, F64, 0, self._hslot)
        elif isinstance(action, torch.Tensor):
            if action.device != self.device or tuple(action.shape) != (self.n_env, self.n_agents, 2):
                raise SwarmError("device action must be [E, N, 2] on the env's device")
            if action.dtype not in (torch.float32, torch.float64):
                action = action.to(torch.float32)
            action = action.contiguous()
            rc = self.lib.swarm_step_host(self.handle, _ptr(action), F64 if action.dtype == torch.float64 else F32, 1, self._hslot)
        else:
            a = np.ascontiguousarray(action)
            if a.dtype not in (np.float32, np.float64):
                a = a.astype(np.float64)
            if a.shape != (2, self.n_env * self.n_agents):
                raise SwarmError("host action must be (2, %d)" % (self.n_env * self.n_agents))
            rc = self.lib.swarm_step_host(self.handle, a.ctypes.data_as(ctypes.c_void_p), F64 if a.dtype == np.float64 else F32, 0,
                                          self._hslot)
        check(self.lib, self.handle, rc)
        return h[self._hslot]

    def llm_action(self):
        """[E, N, 2] float64 device tensor: the 'llm' strategy's action for the current state (assembly.py:525-529)."""
        out = torch.empty((self.n_env, self.n_agents, 2), dtype=torch.float64, device=self.device)
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_get_llm_action(self.handle, _ptr(out)))
        return out

    def indices(self, sensed=True, occupied=True):
        """The reference's index scratch for the current state (debug / parity export)."""
        E, N = self.n_env, self.n_agents
        nei = torch.empty((E, N, self.topo), dtype=torch.int32, device=self.device)
        inf = torch.empty((E, N), dtype=torch.int32, device=self.device)
        sen = torch.empty((E, N, self.g_max), dtype=torch.int32, device=self.device) if sensed else None
        occ = torch.empty((E, N, self.occ_max), dtype=torch.int32, device=self.device) if occupied else None
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_get_indices(self.handle, _ptr(nei), _ptr(inf), _ptr(sen), _ptr(occ)))
        return dict(neighbor_index=nei, in_flags=inf, sensed_index=sen, occupied_index=occ)

    def metrics(self):
        """[E, 3] float64 tensor: coverage_rate, distribution_uniformity, voronoi_based_uniformity per env
        (assembly_wrapper.py:48-128) of the current state."""
        out = torch.empty((self.n_env, 3), dtype=torch.float64, device=self.device)
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_metrics(self.handle, _ptr(out)))
        return out

    def rule_action(self):
        """[E, N, 2] float64 tensor: the rule-based expert's action for the current state (assembly.py:530-601)."""
        out = torch.empty((self.n_env, self.n_agents, 2), dtype=torch.float64, device=self.device)
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_rule_action(self.handle, _ptr(out)))
        return out

    def lattice_envs(self):
        """Number of envs whose target cells were recognised as a lattice subset (fast sensed/occupied path)."""
        return int(self.lib.swarm_lattice_envs(self.handle))

    # -- measurement helpers --------------------------------------------------------------------------
    def algorithmic_bytes_per_step(self):
        return float(self.lib.swarm_step_algorithmic_bytes(self.handle))

    def timer_start(self):
        self._sync_stream()
        check(self.lib, self.handle, self.lib.swarm_timer_start(self.handle))

    def timer_stop(self):
        ms = ctypes.c_float()
        check(self.lib, self.handle, self.lib.swarm_timer_stop(self.handle, ctypes.byref(ms)))
        return float(ms.value)
