#!/usr/bin/env python3
"""bench.py -- agent-steps/sec of the batched AssemblySwarm env step on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched by
torch.distributed.run with one rank per GPU.  A "step" is one fused env step over one batch of synthetic
environments (BASELINE.json: 64 agents x 4096 envs per GPU; weak scaling: 4096 envs on every rank, i.e.
64 x 32768 at 8 GPUs).  Environments are independent, so ranks share nothing on the step path (no RCCL
collective); torch.distributed is used only for the barrier and the max-over-ranks of the timing.

Prints ONE JSON line on rank 0 with the contract's keys plus
  "roofline":     HBM roofline of the step kernel (algorithmic bytes / HIP-event-timed launch duration / 8 TB/s)
  "cpu_baseline": the reference's CPU path timed on this box's host cores on a bounded sample (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(workload):
    """HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, gfx950
    FETCH correction applied) and committed under profiles/: returned only when it was taken on this exact workload,
    otherwise None (it cannot be measured from inside this process)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "final_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if workload.startswith(d.get("_workload", "\0")):
            best = float(d["_traffic_bytes_per_launch"])
    return best


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--agents", type=int, default=64)
    ap.add_argument("--envs", type=int, default=4096, help="environments PER GPU")
    ap.add_argument("--state", choices=["assembled", "scatter"], default="assembled")
    ap.add_argument("--assemble-steps", type=int, default=100)
    ap.add_argument("--seed", type=int, default=226)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU time budget of the baseline sample")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="rehearsal of the N>1 code path on a one-GPU box: every rank uses cuda:0 and the barrier / "
                         "max-over-ranks run over gloo (numbers from such a run are meaningless)")
    return ap.parse_args()


def cpu_baseline(sb, shapes, sy, r_avoid, n_agents, budget_s):
    """Time the reference CPU path on a bounded sample of the SAME workload: the first few environments of
    this batch, from the state the GPU run starts from, advanced with prior-policy actions (what the GPU loop
    does).  kind = "reference": the reference's own libAssemblyEnv.so (compiled unmodified, oracle/_ref) driven
    by a restatement of assembly.py's numpy glue; kind = "port": our plain-C oracle when _ref is not present."""
    from oracle.oracle_py import Oracle, RefLib, ref_step
    p, dp = [x.cpu().numpy() for x in sb.get_state()]
    nei = sb.indices(False, False)["neighbor_index"].cpu().numpy()
    use_ref = RefLib.available()
    ref = RefLib() if use_ref else None
    orc = Oracle()
    n_envs_sample, done_steps = 0, 0
    t_used = 0.0
    steps_per_env = 20
    e = 0
    while t_used < budget_s and e < sb.n_env:
        g = np.ascontiguousarray(sy["cells"][e][:, : sy["n_g"][e]])
        pe, dpe, ne = p[e].copy(), dp[e].copy(), nei[e].copy()
        a = np.zeros((2, n_agents))
        t0 = time.perf_counter()
        for _ in range(steps_per_env):
            if use_ref:
                s = ref_step(ref, pe, dpe, a, g, ne, float(sy["l_cell"][e]), r_avoid)
            else:
                s = orc.step(pe, dpe, a, g, ne, float(sy["l_cell"][e]), r_avoid)
            pe, dpe, ne = s["p"], s["dp"], s["neighbor_index"]
            a = s["a_prior"].astype(np.float32).astype(np.float64)
        t_used += time.perf_counter() - t0
        done_steps += steps_per_env
        n_envs_sample += 1
        e += 1
    value = done_steps * n_agents / t_used
    return {"value": value, "unit": "agent-steps/s", "cores": 1, "kind": "reference" if use_ref else "port",
            "sample": f"{n_envs_sample} envs x {steps_per_env} steps of the same {n_agents}-agent batch "
                      f"(prior-policy actions), {t_used:.1f} s on 1 core; "
                      + ("reference libAssemblyEnv.so + numpy glue" if use_ref else "plain-C oracle port")}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        print("bench.py: no HIP device visible; the env step has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_one_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
    from marl_llm_amd.synth import synthetic_batch

    shapes = synthetic_shape_set()
    n_a, E = args.agents, args.envs
    r_avoid = r_avoid_for(n_a, shapes)
    # weak scaling: every rank owns its own slice [rank*E, (rank+1)*E) of the global env range
    sy = synthetic_batch(E, n_a, shapes, seed=args.seed, env_offset=rank * E)
    sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=r_avoid,
                    device=f"cuda:{local_rank}")
    sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"])
    sb.set_state(sy["p"], sy["dp"])
    sb.observe()
    dev = sb.device
    gen = torch.Generator(device=dev); gen.manual_seed(args.seed + rank)
    if args.state == "assembled":
        act = torch.zeros((E, n_a, 2), dtype=torch.float32, device=dev)
        for _ in range(args.assemble_steps):          # untimed: assemble the swarm with the prior policy
            _, _, _, act = sb.step(act)
    else:
        pool = [torch.rand((E, n_a, 2), generator=gen, device=dev) * 2 - 1 for _ in range(8)]
        act = pool[0]

    def one_step(k, act):
        if args.state == "assembled":
            return sb.step(act)[3]                    # next action := this step's prior (device tensor, no copy)
        sb.step(pool[k % 8])
        return None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(sb, shapes, sy, r_avoid, n_a, args.cpu_seconds)

    for k in range(args.warmup):
        act = one_step(k, act)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    sb.timer_start()                                   # HIP events on the stream the kernel is launched on
    t0 = time.perf_counter()
    for k in range(args.steps):
        act = one_step(k, act)
    kernel_ms = sb.timer_stop()                        # synchronizes on the stop event
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    in_shape = float(sb.indices(False, False)["in_flags"].float().mean().item())
    if rank == 0:
        total_agent_steps = float(world) * E * n_a * args.steps
        value = total_agent_steps / dt
        alg_bytes = sb.algorithmic_bytes_per_step()
        launch_s = kernel_ms * 1e-3 / args.steps
        achieved = alg_bytes / launch_s / 1e9
        workload = (f"assembly env, {n_a} agents x {E} envs per GPU ({n_a} x {E * world} total), assembled state, "
                    f"prior-policy actions" if args.state == "assembled" else
                    f"assembly env, {n_a} agents x {E} envs per GPU ({n_a} x {E * world} total), scatter state, "
                    f"U(-1,1) actions")
        out = {
            "metric": "agent-steps/sec", "value": value, "unit": "agent-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload,
                       "agents": n_a, "envs_per_gpu": E, "envs_total": E * world, "obs_dtype": "f32",
                       "state_dtype": "f64", "in_shape_fraction": round(in_shape, 3), "seed": args.seed,
                       "parallelism": f"env-sharded x{world}, no collective on the step path"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(workload.split(" (")[0] + ", assembled state" if args.state == "assembled" else "-"),
                         "kernel": "k_env<%d,float,true>" % max(8, 1 << (n_a - 1).bit_length()), "kernel_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if cpu is not None:
            cpu["gpu_over_cpu"] = value / cpu["value"]
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    sb.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
