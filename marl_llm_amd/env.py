"""Host-side mirror of the reference's env interface for the step path.

`AssemblySwarmEnv` keeps the surface `marl_llm/train`, `marl_llm/eval` and `AssemblySwarmWrapper` rely on
(/root/reference/cus_gym/gym/envs/customized_envs/assembly.py:15-223,487-666 and
/root/reference/cus_gym/gym/wrappers/customized_envs/assembly_wrapper.py:18-46; the full list is SURVEY.md
section 8b): late configuration through ``__reinit__(args)`` with the same attribute names, ``reset() -> obs``,
``step(a) -> (obs, rew, done, info, a_prior)``, ``n_a / num_agents / agents / agent_types / observation_space /
action_space / alpha / p / dp / grid_center ...``.  Underneath, E independent environments live on one GPU behind
``marl_llm_amd.batched.SwarmBatch``; nothing here computes env arithmetic on the CPU except the reset-time
random draws (host numpy, same draw order as assembly.py:156-219, then uploaded).

Presentation modes
  * ``n_envs == 1``: exactly the reference's shapes -- obs ``(obs_dim, n_a)``, action ``(2, n_a)``.
  * ``n_envs == E > 1``, numpy API: the E environments are presented as ONE env with ``n_a = E * N`` agents on the
    agent axis (env-major), which is what lets the unchanged trainer batch them through its shared policy
    (SURVEY.md section 7.4 item 5).  ``r_avoid`` is derived from the per-environment N (assembly.py:124).
  * tensor API (``reset_tensor`` / ``step_tensor``): device tensors ``[E, N, D]`` -- no host round trip.
"""
import types

import numpy as np

from .shapes import load_results

try:  # the reference vendors a gym 0.19 fork; when it (or any gym) is importable we subclass it
    import gym as _gym
    from gym import spaces as _spaces
    _EnvBase = _gym.Env
    _WrapperBase = _gym.Wrapper
except Exception:  # pragma: no cover - gym absent: minimal stand-ins with the attributes consumers read
    _gym = None
    _spaces = None
    _EnvBase = object
    _WrapperBase = object


class _Box:
    """Stand-in for gym.spaces.Box when gym is not importable (only .shape/.dtype/.low/.high are consumed)."""

    def __init__(self, low, high, shape, dtype):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), np.dtype(dtype)


def _box(shape):
    if _spaces is not None:
        return _spaces.Box(low=-np.inf, high=+np.inf, shape=shape, dtype=np.float32)      # assembly.py:802,806
    return _Box(-np.inf, np.inf, shape, np.float32)


class Agent:                                    # assembly_wrapper.py:5-16
    def __init__(self, adversary=False):
        self.adversary = adversary


class AssemblySwarmEnv(_EnvBase):
    metadata = {"render.modes": ["human", "rgb_array"], "video.frames_per_second": 45}

    def __init__(self, n_envs=1, device="cuda:0", obs_dtype="float64", rng="global", seed=226, host_copy="auto"):
        # constants of assembly.py:18-81
        self.reward_sharing_mode = "individual"
        self.penalize_entering = self.penalize_interaction = self.penalize_exploration = True
        self.dim = 2
        self.n_a = 10
        self.topo_nei_max = 6
        self.act_dim_agent = self.dim
        self.m_a = 1
        self.size_a = 0.035
        self.d_sen = 3
        self.r_avoid = 0.15
        self.Vel_max = 0.8
        self.boundary_width_half = self.boundary_height_half = 2.4
        self.k_ball, self.k_wall, self.c_wall = 30, 100, 5
        self.dt = 0.1
        self.n_frames = 1
        self.sensitivity = 1
        self.simulation_time = 0
        self.n_envs = int(n_envs)
        self._device = device
        self._obs_dtype = obs_dtype
        self._rng_mode = rng
        self._seed = seed
        self._batch = None
        self._cells_dirty = False
        # numpy API: the library hands out views of two pinned host slots that it rewrites every second call.  The
        # reference returns fresh arrays; host_copy=True copies them (always safe), False returns the views (valid until the
        # call after the next -- what the reference's trainer needs, train_assembly.py:97-111), "auto" copies below 4 MB.
        self._host_copy = host_copy
        self._state_version = 0
        self._metrics_cache = (None, None)

    # ------------------------------------------------------------------ configuration (assembly.py:92-154)
    def __reinit__(self, args):
        self.n_agents_per_env = int(args.n_a)
        self.n_a = self.n_agents_per_env * self.n_envs           # presented agent axis
        self.render_traj = getattr(args, "render_traj", False)
        self.traj_len = getattr(args, "traj_len", 15)
        self.is_collected = getattr(args, "is_collected", False)
        self.video = getattr(args, "video", False)
        self.is_boundary = bool(args.is_boundary)
        self.is_periodic = not self.is_boundary
        self.dynamics_mode = args.dynamics_mode
        self.agent_strategy = args.agent_strategy
        self.is_con_self_state = bool(args.is_con_self_state)
        self.is_feature_norm = bool(getattr(args, "is_feature_norm", False))
        self.training_method = args.training_method
        self.alpha = 1
        if self.dynamics_mode != "Cartesian":
            raise ValueError("only dynamics_mode='Cartesian' exists in the reference (assembly.py:141-146)")
        if self.agent_strategy not in ("input", "random", "rule", "llm"):
            raise ValueError("agent_strategy %r: the reference knows 'input', 'random', 'rule' and 'llm' (assembly.py:521-603)" % (self.agent_strategy,))

        results = args.results_file if isinstance(args.results_file, dict) else load_results(args.results_file)
        self.l_cells = list(results["l_cell"])
        self.grid_center_origins = [np.asarray(g, dtype=np.float64) for g in results["grid_coords"]]
        self.binary_images = results.get("binary_image", [None] * len(self.l_cells))
        self.shape_bound_points_origins = results.get("shape_bound_points", [np.zeros(4)] * len(self.l_cells))
        self.num_train_shape = len(self.l_cells)
        self.n_gs = [g.shape[0] for g in self.grid_center_origins]
        self.r_avoid = round(float(np.sqrt(4 * np.min(self.n_gs) / (self.n_agents_per_env * np.pi)) * np.min(self.l_cells)), 2)
        self.num_obs_grid_max = 80
        self.num_occupied_grid_max = 200
        if self._rng_mode == "global":
            np.random.choice([True, False], size=(self.n_agents_per_env, self.n_agents_per_env))   # assembly.py:133 (draws kept)
        self.obs_dim_agent = 2 * self.dim * (self.topo_nei_max + 1 + int(self.is_con_self_state)) + self.dim * self.num_obs_grid_max
        self.observation_space = _box((self.obs_dim_agent, self.n_a))
        self.action_space = _box((self.act_dim_agent, self.n_a))
        self.shape_frequency = np.zeros(len(self.l_cells))
        self.n_cells_max = int(max(self.n_gs))
        self._batch = None
        self._shapes_uploaded = False

    # ------------------------------------------------------------------ backend
    def _backend(self):
        if self._batch is None:
            import torch
            from .batched import SwarmBatch
            dt = torch.float64 if self._obs_dtype in ("float64", "f64") else torch.float32
            self._batch = SwarmBatch(n_env=self.n_envs, n_agents=self.n_agents_per_env, n_cells_max=self.n_cells_max,
                                     r_avoid=self.r_avoid, is_boundary=self.is_boundary,
                                     with_self=self.is_con_self_state,
                                     # 'llm': the agents are driven by the Python twin of the prior (assembly.py:525-529,892-940),
                                     # evaluated on the device next to the prior itself
                                     with_prior=(self.training_method == "llm_rl" or self.agent_strategy == "llm"),
                                     llm_action=(self.agent_strategy == "llm"),
                                     obs_dtype=dt, device=self._device, d_sen=0.4,
                                     topo=self.topo_nei_max, g_max=self.num_obs_grid_max,
                                     occ_max=self.num_occupied_grid_max,
                                     boundary=(-self.boundary_width_half, self.boundary_height_half,
                                               self.boundary_width_half, -self.boundary_height_half),
                                     size_a=self.size_a, k_ball=self.k_ball, k_wall=self.k_wall, c_wall=self.c_wall,
                                     vel_max=self.Vel_max, dt=self.dt)
        return self._batch

    # ------------------------------------------------------------------ reset (assembly.py:156-223)
    def _sample_reset(self):
        """Host-side random draws of reset(), per environment, in the reference's order.  Pure numpy (no GPU)."""
        E, N = self.n_envs, self.n_agents_per_env
        W, H = self.boundary_width_half, self.boundary_height_half
        cells = np.zeros((E, 2, self.n_cells_max)); n_g = np.zeros(E, np.int32); l_cell = np.zeros(E)
        p = np.zeros((E, 2, N)); dp = np.zeros((E, 2, N))
        shape_idx = np.zeros(E, np.int64)
        for e in range(E):
            rs = np.random if self._rng_mode == "global" else np.random.RandomState([self._seed, self._episode, e])
            s = rs.randint(0, self.num_train_shape)                                   # :160
            self.shape_frequency[s] += 1
            origin = self.grid_center_origins[s].T                                    # :164
            ang = np.pi * rs.uniform(-1, 1)                                           # :175
            rot = np.array([[np.cos(ang), np.sin(ang)], [-np.sin(ang), np.cos(ang)]])
            origin = np.dot(rot, 1 * origin)                                          # :170-178 (shape_scale = 1)
            rs.uniform(-1.2, 1.2, (2, 1))                                             # :182 drawn, then discarded
            off = np.array([[rs.uniform(-W + 1, W - 1), rs.uniform(-H + 1, H - 1)]]).T   # :184-185
            g = origin.copy() + off                                                   # :187
            if rs.uniform(-1, 1) > 0:                                                 # :202-208
                pe = np.concatenate((rs.uniform(-W, W, (1, N)), rs.uniform(-H, H, (1, N))), axis=0)
            else:
                pe = rs.uniform(-1, 1, (2, N)) + np.array([[rs.uniform(-W + 1, W - 1), rs.uniform(-H + 1, H - 1)]]).T
            dpe = rs.uniform(-0.5, 0.5, (2, N))                                       # :215
            ng = g.shape[1]
            cells[e, :, :ng] = g; n_g[e] = ng; l_cell[e] = 1 * self.l_cells[s]
            p[e] = pe; dp[e] = dpe; shape_idx[e] = s
        return dict(cells=cells, n_g=n_g, l_cell=l_cell, p=p, dp=dp, shape_index=shape_idx)

    def reset_tensor(self, _observe=True):
        """reset() returning the device observation tensor [E, N, D].

        rng="global" (default): the reference's draw order from numpy's global RNG on the host (seed-for-seed parity,
        assembly.py:156-219).  rng="counter": the same draws from a per-(seed, episode, env) numpy stream on the host.
        rng="device": the batched device-side reset (swarm_reset: counter-based generator on the GPU, no host work)."""
        self.simulation_time = 0
        self._episode = getattr(self, "_episode", -1) + 1
        self.d_sen = 0.4                                                              # :199
        self.boundary_pos = np.array([-self.boundary_width_half, self.boundary_height_half,
                                      self.boundary_width_half, -self.boundary_height_half], dtype=np.float64)
        if self._rng_mode == "device":
            b = self._backend()
            if not getattr(self, "_shapes_uploaded", False):
                b.set_shapes(dict(grid_coords=self.grid_center_origins, l_cell=self.l_cells))
                self._shapes_uploaded = True
            obs = b.reset(self._seed, self._episode, getattr(self, "env_offset", 0))
            self._cells, self._n_g = b.get_cells()
            self.shape_index = b.get_shape_index().astype(np.int64)      # which shape the device drew per env (:160)
            self._l_cell = np.asarray(self.l_cells, dtype=np.float64)[self.shape_index]
            np.add.at(self.shape_frequency, self.shape_index, 1)
            self._cells_dirty = False
            return obs
        s = self._sample_reset()
        self._cells, self._n_g, self._l_cell = s["cells"], s["n_g"], s["l_cell"]
        self.shape_index = s["shape_index"]
        self.d_sen = 0.4                                                              # :199
        self.boundary_pos = np.array([-self.boundary_width_half, self.boundary_height_half,
                                      self.boundary_width_half, -self.boundary_height_half], dtype=np.float64)
        b = self._backend()
        b.set_cells(self._cells, self._n_g, self._l_cell)
        b.set_state(s["p"], s["dp"])
        self._cells_dirty = False
        return b.observe() if _observe else None

    def reset(self):
        self.reset_tensor(_observe=False)         # draws + uploads; the observation pass runs once, in observe_host
        return self._host_obs(self._backend().observe_host())

    def _own(self, a):
        """A host-slot view as the caller gets it: a copy unless host_copy says views are fine."""
        c = self._host_copy
        if c is True or (c == "auto" and a.nbytes <= (4 << 20)):
            return a.copy()
        return a

    def _host_obs(self, obs):
        self._state_version += 1
        return self._own(obs)

    def _flush_cells(self):
        """Upload target cells that were assigned through the attribute setters and refresh the obs-derived caches.

        The reference's eval script switches the target shape by assigning env.env.{l_cell, n_g, grid_center}
        (eval_assembly.py:34-57) and reads the wrapper metrics right after, BEFORE the next step (:154-162); its env
        holds plain attributes, so every reader sees the new cells at once.  Here the cells live on the device: every
        reader of device state (step, metrics, indices, rule action) calls this first."""
        b = self._backend()
        if self._cells_dirty:
            b.set_cells(self._cells, self._n_g, self._l_cell)
            b.observe()
            self._cells_dirty = False
            self._state_version += 1
        return b

    # ------------------------------------------------------------------ step (assembly.py:487-666)
    def _strategy_action(self, b, action):
        """The action the reference's step applies (assembly.py:521-603): the passed one ('input'), a uniform draw from
        numpy's global stream ('random'), the rule-based expert ('rule', device) or the prior's Python twin ('llm': None =
        the library's own device-side action)."""
        if self.agent_strategy == "rule":
            return b.rule_action()
        if self.agent_strategy == "random":
            return np.random.uniform(-1, 1, (self.act_dim_agent, self.n_a))           # assembly.py:523-524
        if self.agent_strategy == "llm":
            return None
        return action

    def step_tensor(self, action):
        """action [E, N, 2] device tensor -> (obs [E,N,D], reward [E,N], done [E,N] uint8, a_prior [E,N,2] | None)."""
        import torch
        b = self._flush_cells()
        self.simulation_time += self.dt
        self._state_version += 1
        action = self._strategy_action(b, action)
        if isinstance(action, np.ndarray):
            E, N = self.n_envs, self.n_agents_per_env
            action = torch.as_tensor(np.ascontiguousarray(action.T.reshape(E, N, 2)), device=b.device)
        applied = b.llm_action() if (action is None and self.is_collected) else action
        obs, rew, done, pri = b.step(action)
        if self.training_method != "llm_rl":
            pri = None
        if self.is_collected:                      # assembly.py:663-664: the applied action is returned instead of the prior
            pri = applied
        return obs, rew, done, pri

    def step(self, a):
        """The reference's numpy API (assembly.py:487-666): a (2, n_a) -> (obs (D, n_a) f64, rew (1, n_a) f64, done (1, n_a)
        bool, info, a_prior (2, n_a) f64 | None | u).  One library call: the action goes up through a pinned staging buffer,
        the outputs come back widened / transposed on the device in one copy into pinned memory (swarm_step_host)."""
        E, N = self.n_envs, self.n_agents_per_env
        a = np.asarray(a)
        if a.shape != (self.act_dim_agent, self.n_a):
            raise ValueError("action must have shape %r" % ((self.act_dim_agent, self.n_a),))
        b = self._flush_cells()
        self.simulation_time += self.dt
        act = self._strategy_action(b, a)
        applied = None
        if self.is_collected:                      # assembly.py:663-664: u, the applied action, is the fifth element
            if act is None:
                applied = np.ascontiguousarray(b.llm_action().reshape(E * N, 2).cpu().numpy().T)
            elif isinstance(act, np.ndarray):
                applied = np.array(act, dtype=np.float64)
            else:
                applied = np.ascontiguousarray(act.reshape(E * N, 2).to("cpu").numpy().astype(np.float64).T)
        out = b.step_host(act)
        obs_np = self._host_obs(out["obs"])
        rew_np, done_np = self._own(out["reward"]), self._own(out["done"])
        info = np.array([None, None, None]).reshape(3, 1)                             # :484-485
        pri_np = self._own(out["a_prior"]) if self.training_method == "llm_rl" else None
        if self.is_collected:
            pri_np = applied
        return obs_np, rew_np, done_np, info, pri_np

    def _obs_to_numpy(self, obs):
        E, N = self.n_envs, self.n_agents_per_env
        return np.ascontiguousarray(obs.reshape(E * N, self.obs_dim_agent).cpu().numpy().astype(np.float64).T)

    # ------------------------------------------------------------------ state the reference exposes as attributes
    @property
    def p(self):
        """(2, n_a) positions, env-major on the agent axis (eval_assembly.py:137,150 reads env.p)."""
        p, _ = self._backend().get_state()
        return np.ascontiguousarray(p.cpu().numpy().transpose(1, 0, 2).reshape(2, -1))

    @property
    def dp(self):
        _, dp = self._backend().get_state()
        return np.ascontiguousarray(dp.cpu().numpy().transpose(1, 0, 2).reshape(2, -1))

    def set_state(self, p, dp):
        """Inject a state ((2, n_a) arrays or [E,2,N]); recomputes the obs-derived caches.  Returns obs."""
        E, N = self.n_envs, self.n_agents_per_env
        p = np.asarray(p, np.float64); dp = np.asarray(dp, np.float64)
        if p.ndim == 2:
            p = p.reshape(2, E, N).transpose(1, 0, 2); dp = dp.reshape(2, E, N).transpose(1, 0, 2)
        b = self._flush_cells() if getattr(self, "_cells", None) is not None else self._backend()
        b.set_state(np.ascontiguousarray(p), np.ascontiguousarray(dp))
        return self._host_obs(b.observe_host())

    def indices(self):
        """neighbor_index / in_flags / sensed_index / occupied_index of the current state (numpy)."""
        return {k: v.cpu().numpy() for k, v in self._flush_cells().indices().items()}

    def metrics_tensor(self):
        """[E, 3] float64 device tensor: coverage_rate, distribution_uniformity, voronoi_based_uniformity per env
        (assembly_wrapper.py:48-128) of the current state and the CURRENT target cells."""
        b = self._flush_cells()
        ver, val = self._metrics_cache
        if ver != self._state_version:             # the three wrapper metrics of one state share one launch + one read-back
            val = b.metrics()
            self._metrics_cache = (self._state_version, val)
        return val

    # grid_center / n_g / l_cell: readable and writable like the reference's attributes (env 0 when E > 1)
    @property
    def grid_center(self):
        return np.ascontiguousarray(self._cells[0][:, : self._n_g[0]])

    @grid_center.setter
    def grid_center(self, g):
        g = np.asarray(g, np.float64)
        if g.shape[1] > self.n_cells_max:
            raise ValueError("grid_center has more cells than n_cells_max=%d" % self.n_cells_max)
        self._cells[:, :, :] = 0.0
        self._cells[:, :, : g.shape[1]] = g
        self._n_g[:] = g.shape[1]
        self._cells_dirty = True

    @property
    def n_g(self):
        return int(self._n_g[0])

    @n_g.setter
    def n_g(self, v):
        self._n_g[:] = int(v); self._cells_dirty = True

    @property
    def l_cell(self):
        return float(self._l_cell[0])

    @l_cell.setter
    def l_cell(self, v):
        self._l_cell[:] = float(v); self._cells_dirty = True

    # ------------------------------------------------------------------ misc gym surface
    def render(self, mode="human"):
        """No-op.  The reference draws the swarm with matplotlib (assembly.py:668-747); drawing is not on the step path,
        but train_assembly.py:94-95 and eval_assembly.py:147 call it inside their loops, so it must exist and return."""
        return None

    def close(self):
        if self._batch is not None:
            self._batch.close()
            self._batch = None

    @property
    def unwrapped(self):
        return self


class AssemblySwarmWrapper(_WrapperBase):
    """assembly_wrapper.py:18-46: calls env.__reinit__(args) and exposes the multi-agent attributes."""

    def __init__(self, env, args):
        if _WrapperBase is not object:
            super().__init__(env)
        else:
            self.env = env
        env.__reinit__(args)
        self.num_agents = self.env.n_a
        self.agents = [Agent() for _ in range(self.num_agents)]
        self.agent_types = ["agent"]
        self.action_space = self.env.action_space
        self.observation_space = self.env.observation_space

    def __getattr__(self, name):                 # gym.Wrapper forwards reads the same way (core.py:232-238)
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    # evaluation metrics (assembly_wrapper.py:48-128), computed on the device; env 0 when several envs are batched
    def _metrics_host(self):
        t = self.env.metrics_tensor()
        if getattr(self, "_mh", (None, None))[0] is not t:
            self._mh = (t, t[0].cpu().numpy())
        return self._mh[1]

    def coverage_rate(self):
        return float(self._metrics_host()[0])

    def distribution_uniformity(self):
        return float(self._metrics_host()[1])

    def voronoi_based_uniformity(self):
        return float(self._metrics_host()[2])

    def render(self, mode="human", **kw):
        return self.env.render(mode=mode, **kw)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, a):
        return self.env.step(a)


def make_args(n_a=30, results_file=None, **over):
    """The env-relevant flags of marl_llm/cfg/assembly_cfg.py:153-168 with their defaults."""
    d = dict(n_a=n_a, is_boundary=True, is_con_self_state=True, is_feature_norm=False, dynamics_mode="Cartesian",
             render_traj=False, traj_len=15, agent_strategy="input", training_method="llm_rl", is_collected=False,
             results_file=results_file, video=False)
    d.update(over)
    return types.SimpleNamespace(**d)


class _Made:
    """What `gym.make(id)` hands back as far as the reference's callers use it: they take `.unwrapped` at once
    (train_assembly.py:49, eval_assembly.py:96; the TimeLimit wrapper gym adds is thereby stripped)."""

    def __init__(self, env):
        self.env = env

    @property
    def unwrapped(self):
        return self.env


def make(env_id="AssemblySwarm-v0", **kw):
    """`gym.make('AssemblySwarm-v0')` for hosts without a gym package (cus_gym/gym/envs/__init__.py:14-19 registers
    exactly this id for the assembly env).  With a gym module present use :func:`register` and gym.make instead."""
    if env_id != "AssemblySwarm-v0":
        raise KeyError("unknown environment id %r (this package provides AssemblySwarm-v0)" % (env_id,))
    return _Made(AssemblySwarmEnv(**kw))


def register(gym_module=None, n_envs=1, **kw):
    """Register 'AssemblySwarm-v0' in a gym registry so `gym.make('AssemblySwarm-v0').unwrapped`
    (train_assembly.py:49) yields this env."""
    g = gym_module or _gym
    if g is None:
        raise RuntimeError("no gym module to register into")
    from gym.envs.registration import register as _reg, registry
    if "AssemblySwarm-v0" in getattr(registry, "env_specs", {}):
        del registry.env_specs["AssemblySwarm-v0"]
    _reg(id="AssemblySwarm-v0", entry_point="marl_llm_amd.env:AssemblySwarmEnv", kwargs=dict(n_envs=n_envs, **kw))
