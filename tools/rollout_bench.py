#!/usr/bin/env python3
"""Device-resident rollout throughput: env step alone vs policy + env vs policy + env + replay push
(SURVEY.md section 8f rank 1).  Everything stays on the GPU; the policy is the reference's actor shape
(192-180-180-180-2 MLP, networks.py:6-44) run by torch (hipBLASLt)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from marl_llm_amd.batched import SwarmBatch
from marl_llm_amd.rollout import ChainedReplay, DeviceReplay, FusedPolicy, PolicyMLP, rollout
from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set


def main():
    n_a = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    E = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    steps = 100
    shapes = synthetic_shape_set()
    ng_max = max(np.asarray(g).shape[0] for g in shapes["grid_coords"])
    sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=ng_max, r_avoid=r_avoid_for(n_a, shapes))
    sb.set_shapes(shapes)
    t0 = time.perf_counter(); obs = sb.reset(seed=226); torch.cuda.synchronize(); t_reset = time.perf_counter() - t0
    t0 = time.perf_counter(); obs = sb.reset(seed=226, episode=1); torch.cuda.synchronize(); t_reset = time.perf_counter() - t0
    policy = PolicyMLP(obs_dim=sb.obs_dim).to(sb.device)
    replay = DeviceReplay(capacity_rows=8 * E * n_a, obs_dim=sb.obs_dim, act_dim=2, device=sb.device)
    act = torch.zeros((E, n_a, 2), device=sb.device)
    for _ in range(50):
        act = sb.step(act)[3]

    def timed(fn):
        fn(5); torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(steps); torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    def env_only(k):
        a = act
        for _ in range(k):
            a = sb.step(a)[3]

    state = {"obs": obs}

    def with_policy(k, rep=None, autocast=False):
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            state["obs"], _ = rollout(sb, policy, k, state["obs"], replay=rep, noise_scale=0.1)

    t_env = timed(env_only)
    t_pol = timed(lambda k: with_policy(k))
    t_pol_bf16 = timed(lambda k: with_policy(k, autocast=True))
    t_full = timed(lambda k: with_policy(k, rep=replay, autocast=True))
    fused = FusedPolicy(policy, device=sb.device)

    def with_fused(k, rep=None):
        state["obs"], _ = rollout(sb, fused, k, state["obs"], replay=rep, noise_scale=0.1)

    def fused_only(k):
        x = state["obs"].reshape(E * n_a, -1)
        for _ in range(k):
            fused(x)

    t_fk = timed(fused_only)
    t_fpol = timed(lambda k: with_fused(k))
    t_ffull = timed(lambda k: with_fused(k, rep=replay))
    n = E * n_a
    print(f"{n_a} agents x {E} envs on one MI355X, device-resident (per step, agent-steps/s):")
    print(f"  batched device reset of all envs        {t_reset * 1e3:8.3f} ms")
    print(f"  env step only (prior-policy actions)    {t_env * 1e3:8.3f} ms   {n / t_env / 1e6:9.1f} M")
    print(f"  + policy MLP fp32 + noise               {t_pol * 1e3:8.3f} ms   {n / t_pol / 1e6:9.1f} M")
    print(f"  + policy MLP bf16 autocast + noise      {t_pol_bf16 * 1e3:8.3f} ms   {n / t_pol_bf16 / 1e6:9.1f} M")
    print(f"  + replay push (obs, act, rew, next_obs) {t_full * 1e3:8.3f} ms   {n / t_full / 1e6:9.1f} M")
    print(f"  fused MFMA policy kernel alone          {t_fk * 1e3:8.3f} ms   {n / t_fk / 1e6:9.1f} M rows/s")
    print(f"  env + fused MFMA policy + noise         {t_fpol * 1e3:8.3f} ms   {n / t_fpol / 1e6:9.1f} M")
    print(f"  + replay push                           {t_ffull * 1e3:8.3f} ms   {n / t_ffull / 1e6:9.1f} M")
    # the same with bfloat16 observation rows end to end (env output, policy input, replay storage)
    sb.close()
    sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=ng_max, r_avoid=r_avoid_for(n_a, shapes), obs_dtype=torch.bfloat16)
    sb.set_shapes(shapes)
    state["obs"] = sb.reset(seed=226)
    a0 = torch.zeros((E, n_a, 2), device=sb.device)
    for _ in range(50):
        state["obs"], _, _, pri = sb.step(a0)
        a0 = pri.float()
    replay16 = DeviceReplay(capacity_rows=8 * E * n_a, obs_dim=sb.obs_dim, act_dim=2, device=sb.device, obs_dtype=torch.bfloat16)

    def env16(k):
        a = a0
        for _ in range(k):
            a = sb.step(a)[3].float()

    t_e16 = timed(env16)
    t_f16 = timed(lambda k: with_fused(k))
    t_f16r = timed(lambda k: with_fused(k, rep=replay16))
    print(f"  bf16 rows: env step only                {t_e16 * 1e3:8.3f} ms   {n / t_e16 / 1e6:9.1f} M")
    print(f"  bf16 rows: env + fused policy + noise   {t_f16 * 1e3:8.3f} ms   {n / t_f16 / 1e6:9.1f} M")
    print(f"  bf16 rows: + replay push                {t_f16r * 1e3:8.3f} ms   {n / t_f16r / 1e6:9.1f} M")
    chain16 = ChainedReplay(8, E * n_a, sb.obs_dim, 2, sb.device, obs_dtype=torch.bfloat16)
    t_f16c = timed(lambda k: with_fused(k, rep=chain16))
    print(f"  bf16 rows: + chained replay push        {t_f16c * 1e3:8.3f} ms   {n / t_f16c / 1e6:9.1f} M")

    def fused_ring(k):          # the fused path: in-kernel noise, transition written in place, no per-step reward reduction
        state["obs"], _ = rollout(sb, fused, k, state["obs"], replay=chain16, noise_scale=0.1, track_reward=False)
    t_ring16 = timed(fused_ring)
    print(f"  bf16 rows: fused ring (2 launches/step) {t_ring16 * 1e3:8.3f} ms   {n / t_ring16 / 1e6:9.1f} M")
    sb.close()


if __name__ == "__main__":
    main()
