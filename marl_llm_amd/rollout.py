"""Device-resident rollout loop.

The reference rolls out on the host: `torch.Tensor(obs)` -> policy MLP on the CPU -> numpy actions -> `env.step` -> numpy
ring buffer (train_assembly.py:91-111, maddpg.py:72-87, agents.py:69-96, buffer_agent.py:67-128).  Once the env emits
`[E, N, D]` device tensors that round trip is the bottleneck, so this module keeps the whole transition on the GPU:

* `PolicyMLP`     -- the reference's actor shape (networks.py:6-44: 4 x Linear, leaky-ReLU, tanh out) as a plain torch
                     module (the learner trains this one).
* `FusedPolicy`   -- the same forward as one hand-written bf16-MFMA kernel (csrc/policy_mlp.hip) for the rollout.
* `DeviceReplay`  -- the ring buffer of buffer_agent.py:13-128 with one row per (env, agent) transition, as device tensors.
* `ChainedReplay` -- the same transitions stored as a ring of env steps that share observation rows (half the copy per push).
* `rollout`       -- obs -> policy -> exploration noise (agents.py:82-96 continuous branch) -> env.step_tensor -> push.

PyTorch is plumbing here (device memory, GEMMs); the environment step is the HIP library.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class PolicyMLP(nn.Module):
    def __init__(self, obs_dim=192, act_dim=2, hidden_dim=180):          # assembly_cfg.py:185 hidden_dim = 180
        super().__init__()
        self.fc1 = nn.Linear(obs_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.fc3 = nn.Linear(hidden_dim, hidden_dim)
        self.fc4 = nn.Linear(hidden_dim, act_dim)

    def forward(self, x):
        h = F.leaky_relu(self.fc1(x))
        h = F.leaky_relu(self.fc2(h))
        h = F.leaky_relu(self.fc3(h))
        return torch.tanh(self.fc4(h))


class FusedPolicy:
    """The same actor evaluated by ONE hand-written HIP kernel (csrc/policy_mlp.hip: bf16 MFMA, fp32 accumulate, the four
    layers chained through the accumulator registers) instead of four hipBLASLt GEMMs + elementwise kernels.  Inference
    only (rollouts); numerics = torch.autocast(bfloat16) on the module.  Built from a PolicyMLP (or any module with
    fc1..fc4); call `refresh()` after the learner updated the weights.  No CPU path."""

    def __init__(self, module, device="cuda:0", precision="bf16"):
        """precision: "bf16" (operands rounded to bfloat16: torch.autocast's contract, ~4e-2 from the fp32 module) or "bf16x3"
        (high + low bfloat16 parts, three MFMAs per product: ~1e-4 from the fp32 module, the reference's actor arithmetic
        for rollouts that must follow networks.py:6-44 closely)."""
        import ctypes
        if precision not in ("bf16", "bf16x3"):
            raise ValueError("precision must be 'bf16' or 'bf16x3'")
        self.precision = precision
        from . import _lib
        self._ctypes = ctypes
        self.lib = _lib.load()
        self.module = module
        dev = torch.device(device)
        if dev.type != "cuda":
            raise ValueError("FusedPolicy needs a cuda (HIP) device; there is no CPU path")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.handle = None
        self.refresh()

    def refresh(self):
        ct = self._ctypes
        m = self.module
        ws = [t.detach().to("cpu", torch.float32).contiguous() for t in
              (m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, m.fc3.weight, m.fc3.bias, m.fc4.weight, m.fc4.bias)]
        self.in_dim, self.hidden, self.act_dim = ws[0].shape[1], ws[0].shape[0], ws[6].shape[0]
        h = ct.c_void_p()
        rc = self.lib.swarm_policy_create(*[ct.c_void_p(w.data_ptr()) for w in ws], self.in_dim, self.hidden, self.act_dim,
                                          self.device.index, ct.byref(h))
        if rc != 0:
            raise RuntimeError("swarm_policy_create failed: " + self.lib.swarm_policy_last_error().decode())
        self.close()
        self.handle = h
        if self.lib.swarm_policy_set_precision(self.handle, 1 if self.precision == "bf16x3" else 0) != 0:
            raise RuntimeError("swarm_policy_set_precision failed: " + self.lib.swarm_policy_last_error().decode())

    def __call__(self, obs, out=None, noise_scale=0.0, seed=0, step=0):
        """obs [rows, in_dim] float32 or bfloat16 on the device (contiguous) -> actions [rows, act_dim] float32.
        noise_scale > 0: the exploring actor of agents.py:93-96 in the same launch -- clamp(action + noise_scale * N(0, 1),
        -1, 1) with a counter-based generator keyed by (seed, step, row).  out: where to write (a contiguous float32
        tensor of rows * act_dim elements, e.g. a replay-ring slot)."""
        if (obs.dtype not in (torch.float32, torch.bfloat16) or not obs.is_contiguous() or obs.device != self.device
                or obs.shape[-1] != self.in_dim):
            raise ValueError("FusedPolicy expects a contiguous float32 / bfloat16 [rows, %d] tensor on %s" % (self.in_dim, self.device))
        rows = obs.numel() // self.in_dim
        if out is None:
            out = torch.empty((rows, self.act_dim), dtype=torch.float32, device=self.device)
        elif (out.dtype != torch.float32 or out.device != self.device or not out.is_contiguous() or out.numel() != rows * self.act_dim):
            raise ValueError("out must be a contiguous float32 tensor of %d elements on %s" % (rows * self.act_dim, self.device))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.swarm_policy_forward_explore(self.handle, obs.data_ptr(), int(obs.dtype == torch.bfloat16), rows, out.data_ptr(),
                                                   float(noise_scale), int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1), stream)
        if rc != 0:
            raise RuntimeError("swarm_policy_forward failed: " + self.lib.swarm_policy_last_error().decode())
        return out.view(rows, self.act_dim)

    def close(self):
        if self.handle is not None:
            self.lib.swarm_policy_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceReplay:
    """Ring buffer of per-agent transitions on the device (buffer_agent.py:40-128 semantics: rows are appended in
    blocks of one env step = E*N rows; a block that would overflow the end is written flush with the end)."""

    def __init__(self, capacity_rows, obs_dim, act_dim, device, obs_dtype=torch.float32):
        self.capacity = int(capacity_rows)
        z = lambda d, dt=torch.float32: torch.zeros((self.capacity, d), dtype=dt, device=device)
        self.obs, self.next_obs = z(obs_dim, obs_dtype), z(obs_dim, obs_dtype)
        self.act, self.act_prior = z(act_dim), z(act_dim)
        self.rew, self.done = z(1), z(1)
        self.filled_i = 0
        self.curr_i = 0

    def __len__(self):
        return self.filled_i

    def push(self, obs, act, rew, next_obs, done, act_prior=None):
        """obs/next_obs [E,N,D], act/act_prior [E,N,2], rew/done [E,N]."""
        n = obs.shape[0] * obs.shape[1]
        if n > self.capacity:
            raise ValueError("one env step (%d rows) does not fit in the replay buffer (%d rows)" % (n, self.capacity))
        if self.curr_i + n > self.capacity:                       # buffer_agent.py:97-100
            self.curr_i = self.capacity - n
        s = slice(self.curr_i, self.curr_i + n)
        self.obs[s] = obs.reshape(n, -1); self.next_obs[s] = next_obs.reshape(n, -1)
        self.act[s] = act.reshape(n, -1)
        self.rew[s] = rew.reshape(n, 1); self.done[s] = done.reshape(n, 1).to(self.done.dtype)
        if act_prior is not None:
            self.act_prior[s] = act_prior.reshape(n, -1)
        self.curr_i += n
        if self.filled_i < self.capacity:                         # buffer_agent.py:122-123: the count may overshoot the
            self.filled_i += n                                    # capacity by up to one block, exactly like the reference's
        if self.curr_i == self.capacity:
            self.curr_i = 0

    def sample(self, batch, generator=None):
        """Uniform with replacement over the rows written so far.  NOT the reference's rule (see sample_reference): that
        one ignores how much of the buffer is filled and hands out never-written zero rows early in training."""
        idx = torch.randint(0, min(self.filled_i, self.capacity), (batch,), device=self.obs.device, generator=generator)
        return self.obs[idx], self.act[idx], self.rew[idx], self.next_obs[idx], self.done[idx], self.act_prior[idx]

    def sample_reference(self, batch, generator=None, begin_index_range=300000):
        """buffer_agent.py:145-154 as is: `batch` DISTINCT rows from the window
        [begin, capacity - begin_index_range + begin), begin ~ U{0 .. begin_index_range - 1} -- a sliding window over the
        whole allocation, whatever has been written.  Needs capacity > begin_index_range + batch (the reference allocates
        2e4 steps x n_agents rows, train_assembly.py:66-69)."""
        R = int(begin_index_range)
        width = self.capacity - R
        if width < batch:
            raise ValueError("sample_reference: capacity %d leaves a window of %d rows for a batch of %d" % (self.capacity, width, batch))
        dev = self.obs.device
        begin = int(torch.randint(0, R, (1,), generator=generator, device=dev).item())
        idx = torch.randperm(width, device=dev, generator=generator)[:batch] + begin
        return self.obs[idx], self.act[idx], self.rew[idx], self.next_obs[idx], self.done[idx], self.act_prior[idx]


class ChainedReplay:
    """Replay of whole env steps in which consecutive transitions SHARE their observation rows: a ring of S = K + 1 step
    slots; transition j is (obs slot j, act/rew/done/prior of slot j, next_obs = obs slot j + 1).  A push therefore copies
    one observation block instead of two (the observations are ~97 % of a transition's bytes).  The K most recent steps
    are valid; the slot after the newest holds that step's next_obs and is excluded from sampling.  Same transitions as
    DeviceReplay / the reference's buffer (buffer_agent.py:67-128), different storage.  Call `break_chain()` when the
    next pushed obs is NOT the previous next_obs (after an env reset)."""

    def __init__(self, n_steps, rows_per_step, obs_dim, act_dim, device, obs_dtype=torch.float32):
        self.K, self.S, self.n = int(n_steps), int(n_steps) + 1, int(rows_per_step)
        z = lambda d, dt=torch.float32: torch.zeros((self.S, self.n, d), dtype=dt, device=device)
        self.obs = z(obs_dim, obs_dtype)
        # act_prior in the env's output dtype and done as the env's uint8, so that a step can write them in place
        self.act, self.act_prior = z(act_dim), z(act_dim, obs_dtype)
        self.rew, self.done = z(1), z(1, torch.uint8)
        self.cur, self.count, self._chained = 0, 0, False

    # zero-copy use (rollout's fused path): the policy writes its action and the env step its outputs straight into the slots
    def begin_step(self, obs):
        """Slots of the transition about to be taken: dict(obs_in = observation rows of the current slot (copied from
        `obs` unless the chain already holds them), act, rew, done, prior, next_obs).  Follow with end_step()."""
        c, nx = self.cur, (self.cur + 1) % self.S
        if not self._chained:
            self.obs[c].copy_(obs.reshape(self.n, -1))
        return dict(obs_in=self.obs[c], act=self.act[c], rew=self.rew[c], done=self.done[c], prior=self.act_prior[c],
                    next_obs=self.obs[nx])

    def end_step(self):
        self.cur, self.count, self._chained = (self.cur + 1) % self.S, min(self.count + 1, self.K), True

    def __len__(self):
        return self.count * self.n

    def break_chain(self):
        self._chained = False

    def push(self, obs, act, rew, next_obs, done, act_prior=None):
        n = self.n
        if obs.shape[0] * obs.shape[1] != n:
            raise ValueError("ChainedReplay takes whole env steps of %d rows" % n)
        c, nx = self.cur, (self.cur + 1) % self.S
        if not self._chained:
            self.obs[c] = obs.reshape(n, -1)
        self.obs[nx] = next_obs.reshape(n, -1)
        self.act[c] = act.reshape(n, -1)
        self.rew[c] = rew.reshape(n, 1); self.done[c] = done.reshape(n, 1).to(self.done.dtype)
        if act_prior is not None:
            self.act_prior[c] = act_prior.reshape(n, -1).to(self.act_prior.dtype)
        self.cur, self.count, self._chained = nx, min(self.count + 1, self.K), True

    def sample(self, batch, generator=None):
        dev = self.obs.device
        back = torch.randint(0, self.count, (batch,), device=dev, generator=generator)
        j = (self.cur - 1 - back) % self.S
        r = torch.randint(0, self.n, (batch,), device=dev, generator=generator)
        jn = (j + 1) % self.S
        return (self.obs[j, r], self.act[j, r], self.rew[j, r], self.obs[jn, r], self.done[j, r].to(torch.float32),
                self.act_prior[j, r].to(torch.float32))


@torch.no_grad()
def rollout(env, policy, steps, obs, replay=None, noise_scale=0.0, epsilon=0.0, generator=None, host_rng=None,
            track_reward=True, seed=0, step0=0):
    """Run `steps` env steps entirely on the device.

    env   : object with step_tensor(action[E,N,2]) -> (obs[E,N,D], rew[E,N], done[E,N], a_prior[E,N,2]|None)
            (marl_llm_amd.env.AssemblySwarmEnv, or a SwarmBatch through `step`)
    obs   : current observation tensor [E,N,D] (from reset_tensor / the previous rollout)
    The epsilon coin of agents.py:89 is drawn on the HOST (numpy, like the reference's np.random.rand()): a device-side
    draw would cost a host synchronisation every step.  `host_rng`: anything with a .random() method -- the np.random
    module (default), a RandomState or a Generator (np.random.default_rng).

    Fused path (policy is a FusedPolicy, replay a ChainedReplay, env a SwarmBatch): TWO launches per step and no copies --
    the policy kernel adds the exploration noise in its epilogue (counter-based generator keyed by (seed, step0 + t, row))
    and writes the action into the replay slot; the env step writes next_obs / reward / done / prior into the ring.
    Everywhere else: the policy's action + torch noise, then `replay.push`.
    track_reward: also return the mean reward of every step ([steps] tensor; one small reduction per step).
    Returns (last obs, mean reward per step tensor [steps] or None)."""
    import numpy as np
    from .batched import SwarmBatch
    step = env.step_tensor if hasattr(env, "step_tensor") else env.step
    E, N, D = obs.shape
    rews = torch.zeros(steps, device=obs.device) if track_reward else None
    coin = host_rng if host_rng is not None else np.random
    fused = isinstance(policy, FusedPolicy) and isinstance(replay, ChainedReplay) and isinstance(env, SwarmBatch)
    for t in range(steps):
        explore_uniform = epsilon > 0 and coin.random() < epsilon                                     # agents.py:89-91
        if fused:
            sl = replay.begin_step(obs)
            if explore_uniform:
                sl["act"].copy_(torch.rand((E * N, policy.act_dim), device=obs.device, generator=generator) * 2 - 1)
            else:
                policy(sl["obs_in"], out=sl["act"], noise_scale=noise_scale, seed=seed, step=step0 + t)
            next_obs, rew, done, pri = env.step(sl["act"].view(E, N, 2),
                                                out=dict(obs=sl["next_obs"], rew=sl["rew"], done=sl["done"], prior=sl["prior"]))
            replay.end_step()
        else:
            x = obs.reshape(E * N, D)
            if x.dtype != torch.float32 and not (x.dtype == torch.bfloat16 and isinstance(policy, FusedPolicy)):
                x = x.float()
            if explore_uniform:
                act = torch.rand((E * N, 2), device=obs.device, generator=generator) * 2 - 1
            elif isinstance(policy, FusedPolicy):
                act = policy(x, noise_scale=noise_scale, seed=seed, step=step0 + t)                   # noise in the kernel's epilogue
            else:
                act = policy(x)
                if noise_scale > 0:                                                                    # agents.py:93-96
                    act = (act + noise_scale * torch.randn(act.shape, device=obs.device, generator=generator)).clamp_(-1, 1)
            act = act.reshape(E, N, 2)
            next_obs, rew, done, pri = step(act)
            if replay is not None:
                replay.push(obs, act, rew, next_obs, done, pri)
        if track_reward:
            rews[t] = rew.mean()
        obs = next_obs
    return obs, rews
