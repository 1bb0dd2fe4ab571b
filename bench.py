#!/usr/bin/env python3
"""bench.py -- agent-steps/sec of the batched AssemblySwarm env step on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no torchrun environment this
process SPAWNS the N ranks itself (fresh interpreters, before anything here touches the GPU; one LOCAL_RANK each);
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` it is one of the N ranks.  Either way
`WORLD_SIZE` must equal `--gpus` or the run fails loudly.  A "step" is one fused env step over one batch of synthetic
environments (BASELINE.json: 64 agents x 4096 envs per GPU; weak scaling: 4096 envs on every rank, i.e. 64 x 32768 at
8 GPUs).  Environments are independent, so ranks share nothing on the step path (no RCCL collective);
torch.distributed (through marl_llm_amd.dist_util) only provides the barrier and the max-over-ranks of the timing.

Prints ONE JSON line on rank 0 with the contract's keys plus
  "roofline":      HBM roofline of the step kernel: SURVEY section 8(d) algorithmic bytes / HIP-event-timed launch
                   duration / 8 TB/s
  "cpu_baseline":  the reference's CPU path (oracle/_ref: its own C++ compiled unmodified + a restatement of its numpy
                   glue) timed on this box's host cores on a bounded sample: all cores (independent worker processes
                   over envs, which is how the reference would use a host) and one core
  "other_configs": kernel time + roofline fraction of the other shapes BASELINE.json names (N = 1 only)
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
KERNEL_SRC = os.path.join(ROOT, "marl_llm_amd", "csrc", "swarm_env.hip")


def source_hash():
    return hashlib.sha256(open(KERNEL_SRC, "rb").read()).hexdigest()[:16]


def pmc_traffic(workload_key):
    """HBM bytes per launch measured with rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs, gfx950
    FETCH correction applied) and committed under profiles/ (tools/collect_profiles.sh).  Counters cannot be read from
    inside this process, so the figure is returned only when the summary was taken on this exact workload AND on this
    exact kernel source (sha256 of csrc/swarm_env.hip stored in the summary); otherwise None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "final_pmc_summary.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("_workload") == workload_key and d.get("_source_sha256") == source_hash():
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
    ap.add_argument("--no-other-configs", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="wall budget of each CPU baseline leg")
    ap.add_argument("--cpu-workers", type=int, default=0, help="worker processes of the all-cores leg (0 = host cores)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="rehearsal of the N>1 code path on a one-GPU box: every rank uses cuda:0 and the barrier / "
                         "max-over-ranks run over gloo (numbers from such a run are meaningless)")
    ap.add_argument("--dry-spawn", action="store_true",
                    help="N>1 plumbing check without a GPU: ranks join gloo, report rank / world, exit")
    ap.add_argument("--dry-fail-rank", type=int, default=-1, help="with --dry-spawn: this rank exits 3 at once (fail-fast test)")
    ap.add_argument("--prewarm-ms", type=float, default=60.0,
                    help="untimed steps of the same loop issued right before --warmup until this much GPU time has passed")
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--latency-worker", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
# N > 1 without torchrun: spawn the ranks
# ----------------------------------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv=None):
    """Start n fresh interpreters running this script as ranks 0..n-1 (children of this process, which never touches
    the GPU; nothing is re-exec'd).  Rank 0 inherits stdout, so its ONE JSON line is this command's output; the other
    ranks' stdout goes to stderr (their failures must be readable).  All children are polled together: the first
    non-zero exit kills the rest at once instead of leaving rank 0 blocked in a rendezvous until a timeout."""
    port = int(os.environ.get("MASTER_PORT", 0)) or free_port()
    argv = sys.argv[1:] if argv is None else argv
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    deadline = time.time() + 1500
    live = set(range(n))
    while live and rc == 0:
        for r in sorted(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code
                    print(f"bench.py: rank {r} failed (exit code {code}); stopping the other ranks", file=sys.stderr)
                    break
        if time.time() > deadline:
            rc = 124
            print("bench.py: ranks still running at the 1500 s deadline", file=sys.stderr)
        if live and rc == 0:
            time.sleep(0.2)
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                pass
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference's CPU path on the host cores
# ----------------------------------------------------------------------------------------------------------------------
def host_cores():
    """Cores this process may actually use: scheduler affinity, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


def cpu_worker_main():
    """One worker of the CPU baseline (started BEFORE the parent initialises the GPU; blocks on stdin until the parent
    has the assembled state).  Steps its own slice of the batch's environments, one env at a time, `steps_per_env`
    steps each with prior-policy actions, until the wall budget is spent.  Prints one JSON line."""
    line = sys.stdin.readline()
    if not line.strip():
        return 0
    job = json.loads(line)
    from oracle.oracle_py import Oracle, RefLib, ref_step
    z = np.load(job["state"])
    use_ref = RefLib.available()
    ref = RefLib() if use_ref else None
    orc = Oracle()
    n_a, ra, spe = int(job["n_agents"]), float(job["r_avoid"]), int(job["steps_per_env"])
    w, nw = int(job["worker"]), int(job["workers"])
    E = z["p"].shape[0]
    done_steps, n_envs = 0, 0
    t0 = time.perf_counter()
    e = w
    while time.perf_counter() - t0 < job["budget_s"] and e < E:
        g = np.ascontiguousarray(z["cells"][e][:, : z["n_g"][e]])
        pe, dpe, ne = z["p"][e].copy(), z["dp"][e].copy(), z["nei"][e].copy()
        a = np.zeros((2, n_a))
        for _ in range(spe):
            if use_ref:
                s = ref_step(ref, pe, dpe, a, g, ne, float(z["l_cell"][e]), ra)
            else:
                s = orc.step(pe, dpe, a, g, ne, float(z["l_cell"][e]), ra)
            pe, dpe, ne = s["p"], s["dp"], s["neighbor_index"]
            a = s["a_prior"].astype(np.float32).astype(np.float64)
        done_steps += spe; n_envs += 1
        e += nw
    print(json.dumps({"worker": w, "env_steps": done_steps, "envs": n_envs, "seconds": time.perf_counter() - t0,
                      "kind": "reference" if use_ref else "port"}), flush=True)
    return 0


def start_cpu_workers(n):
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    return [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker"], env=env, stdin=subprocess.PIPE,
                             stdout=subprocess.PIPE, text=True) for _ in range(n)]


def run_cpu_leg(workers, state_path, n_agents, r_avoid, budget_s, steps_per_env=20):
    nw = len(workers)
    for w, p in enumerate(workers):
        p.stdin.write(json.dumps({"state": state_path, "n_agents": n_agents, "r_avoid": r_avoid, "budget_s": budget_s,
                                  "steps_per_env": steps_per_env, "worker": w, "workers": nw}) + "\n")
        p.stdin.flush()
    res = []
    for p in workers:
        out, _ = p.communicate(timeout=budget_s * 6 + 120)
        res.append(json.loads(out.strip().splitlines()[-1]))
    wall = max(r["seconds"] for r in res)
    steps = sum(r["env_steps"] for r in res)
    return {"value": steps * n_agents / wall, "envs": sum(r["envs"] for r in res), "seconds": wall, "kind": res[0]["kind"]}


def capture_cpu_state(sb, sy):
    """Save the state the GPU run is about to start from (positions, velocities, neighbour lists, cells) for the CPU legs."""
    p, dp = [x.cpu().numpy() for x in sb.get_state()]
    nei = sb.indices(False, False)["neighbor_index"].cpu().numpy()
    fd, path = tempfile.mkstemp(suffix=".npz", prefix="bench_cpu_")
    os.close(fd)
    np.savez(path, p=p, dp=dp, nei=nei, cells=sy["cells"], n_g=sy["n_g"], l_cell=sy["l_cell"])
    return path


def cpu_baseline(path, r_avoid, n_agents, pool_all, pool_one, budget_s):
    """Time the reference CPU path on a bounded sample of the SAME workload: environments of this batch, from the state
    the GPU run starts from (`path`, capture_cpu_state), advanced with prior-policy actions (what the GPU loop does).
    kind = "reference": the reference's own libAssemblyEnv.so (compiled unmodified, oracle/_ref) driven by a restatement
    of assembly.py's numpy glue; kind = "port": our plain-C oracle when _ref is not present.  Two legs: one worker on
    one core, then P independent worker processes (P = host cores), each stepping its own envs -- the reference itself
    is single-threaded (c_lib.py:40-41 sets OMP_NUM_THREADS but the C++ has no pragma), so independent processes over
    envs are how it would use a whole host."""
    try:
        one = run_cpu_leg(pool_one, path, n_agents, r_avoid, min(budget_s, 6.0))
        allc = run_cpu_leg(pool_all, path, n_agents, r_avoid, budget_s)
    finally:
        os.unlink(path)
    P = len(pool_all)
    what = "reference libAssemblyEnv.so + numpy glue" if allc["kind"] == "reference" else "plain-C oracle port"
    return {"value": allc["value"], "unit": "agent-steps/s", "cores": P, "kind": allc["kind"],
            "sample": f"{allc['envs']} envs x 20 steps of the same {n_agents}-agent batch (prior-policy actions), "
                      f"{P} independent worker processes for {allc['seconds']:.1f} s; {what}",
            "one_core": {"value": one["value"], "cores": 1,
                         "sample": f"{one['envs']} envs x 20 steps, {one['seconds']:.1f} s on 1 core"}}


# ----------------------------------------------------------------------------------------------------------------------
# the GPU measurement
# ----------------------------------------------------------------------------------------------------------------------
def survey_bytes(n_agents, n_g):
    """B = 821 B per agent-step (action 8 r, p/dp 32 r+w as fp32, obs 768 w, reward 4 + done 1 +
    prior 8 w) + the env's grid_center read once as fp32 (2 * n_g * 4 B), summed over the batch."""
    return float(len(n_g)) * n_agents * 821.0 + 8.0 * float(np.sum(n_g))


def measure(torch, n_a, E, state, steps, warmup, assemble_steps, seed, env_offset, device, barrier=None, shapes=None):
    """Set up E envs of n_a agents on `device`, bring them to `state`, return a closure that times `steps` steps (and the
    SwarmBatch + inputs for the CPU baseline).  shapes: a results.pkl-layout dict (default: the synthetic shape set)."""
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
    from marl_llm_amd.synth import synthetic_batch
    shapes = synthetic_shape_set() if shapes is None else shapes
    r_avoid = r_avoid_for(n_a, shapes)
    sy = synthetic_batch(E, n_a, shapes, seed=seed, env_offset=env_offset)
    extra = {}
    if os.environ.get("SWARM_BENCH_GMAX"):                 # diagnostic: another list length (LDS footprint / occupancy experiments)
        extra["g_max"] = int(os.environ["SWARM_BENCH_GMAX"])
    sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=sy["cells"].shape[2], r_avoid=r_avoid, device=device, **extra)
    sb.set_cells(sy["cells"], sy["n_g"], sy["l_cell"])
    sb.set_state(sy["p"], sy["dp"])
    sb.observe()
    dev = sb.device
    gen = torch.Generator(device=dev); gen.manual_seed(seed + env_offset)
    if state == "assembled":
        act = torch.zeros((E, n_a, 2), dtype=torch.float32, device=dev)
        for _ in range(assemble_steps):               # untimed: assemble the swarm with the prior policy
            _, _, _, act = sb.step(act)
        pool = None
    else:
        pool = [torch.rand((E, n_a, 2), generator=gen, device=dev) * 2 - 1 for _ in range(8)]
        act = pool[0]

    def one_step(k, act):
        if state == "assembled":
            return sb.step(act)[3]                    # next action := this step's prior (device tensor, no copy)
        sb.step(pool[k % 8])
        return None

    def timed():
        nonlocal act
        for k in range(warmup):
            act = one_step(k, act)
        torch.cuda.synchronize(dev)
        if barrier is not None:
            barrier()
        torch.cuda.synchronize(dev)
        sb.timer_start()                               # HIP events on the stream the kernel is launched on
        t0 = time.perf_counter()
        for k in range(steps):
            act = one_step(k, act)
        kernel_ms = sb.timer_stop()                    # synchronizes on the stop event
        torch.cuda.synchronize(dev)
        if barrier is not None:
            barrier()
        return time.perf_counter() - t0, kernel_ms

    def prewarm(ms):
        """Untimed steps of the same loop until `ms` of GPU time have passed (at least 20): the timed region then starts
        at the clocks the chip holds under this load, whatever the host did before."""
        nonlocal act
        n, k = 0, 0
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        while n < 20 or (time.perf_counter() - t0) * 1e3 < ms:
            for _ in range(20):
                act = one_step(k, act); k += 1
            n += 20
            torch.cuda.synchronize(dev)
        return n

    return sb, sy, r_avoid, timed, prewarm


def latency_worker_main():
    """Child process (part of the CPU-baseline leg: it loads oracle/): per-step latency of the REFERENCE at its own
    scale -- one env, its numpy glue (restated: oracle_py.ref_step) around the five native calls -- with the calls going
    (a) to the reference's own libAssemblyEnv.so (oracle/_ref) and (b) to this library's legacy symbols, the same
    signatures served by HIP kernels (BASELINE config 0: '8 agents x 1 env, plumbing').  Prints one JSON line."""
    from oracle.oracle_py import Oracle, RefLib, ref_step
    from marl_llm_amd import _lib
    from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
    from marl_llm_amd.synth import synthetic_batch
    shapes = synthetic_shape_set()
    out = {}
    ours = RefLib.__new__(RefLib)
    ours.lib = _lib.load()
    have_ref = RefLib.available()
    ref = RefLib() if have_ref else None
    orc = Oracle()
    for n_a in (8, 30):
        ra = r_avoid_for(n_a, shapes)
        sy = synthetic_batch(1, n_a, shapes, seed=226, assembled_fraction=0.6)
        g = np.ascontiguousarray(sy["cells"][0][:, : sy["n_g"][0]])
        l_cell = float(sy["l_cell"][0])
        nei0 = orc.get_observation(sy["p"][0], sy["dp"][0], g, l_cell, ra)["neighbor_index"]
        for name, lib in (("reference_cpu", ref), ("legacy_shim_gpu", ours)):
            if lib is None:
                continue
            p, dp, nei, a = sy["p"][0].copy(), sy["dp"][0].copy(), nei0.copy(), np.zeros((2, n_a))
            for k in range(230):
                if k == 30:
                    t0 = time.perf_counter()
                s_ = ref_step(lib, p, dp, a, g, nei, l_cell, ra)
                p, dp, nei = s_["p"], s_["dp"], s_["neighbor_index"]
                a = s_["a_prior"].astype(np.float32).astype(np.float64)
            out[f"{name}_n{n_a}_ms_per_step"] = (time.perf_counter() - t0) / 200 * 1e3
    print(json.dumps(out), flush=True)
    return 0


def extra_configs(torch, args, device):
    """other_configs beyond the kernel shapes: the numpy drop-in API (the path the unchanged trainer drives,
    train_assembly.py:97-111) at the reference's default scale and at the headline batch, the legacy shim, the device
    rollout.  Every part reports its own failure instead of taking the bench line down."""
    from marl_llm_amd.env import AssemblySwarmEnv, make_args
    from marl_llm_amd.shapes import synthetic_shape_set
    res = []
    shapes = synthetic_shape_set()
    # (1) reference-scale latency of the reference itself and of the legacy shim (child process: it may load oracle/)
    lat = {}
    try:
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--latency-worker"], capture_output=True, text=True,
                           timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"))
        lat = json.loads(o.stdout.strip().splitlines()[-1])
    except Exception as ex:                                            # reported, never fatal for the bench line
        lat = {"error": repr(ex)[:200]}
    # (2) numpy API, 1 env x 30 agents (assembly_cfg.py:153 default): obs / reward / prior come back as the reference's
    #     float64 host arrays every step
    for n_envs, n_a, steps in ((1, 30, 300), (4096, 64, 12)):
        label = (f"numpy drop-in API (AssemblySwarmEnv.step), {n_a} agents x {n_envs} env(s), float64 host arrays in the "
                 "reference's layouts")
        try:
            env = AssemblySwarmEnv(n_envs=n_envs, device=device, obs_dtype="float64", rng="counter", seed=args.seed, host_copy=False)
            env.__reinit__(make_args(n_a=n_a, results_file=shapes))
            env.reset()
            a = np.zeros((2, n_envs * n_a), np.float32)
            for _ in range(5 if n_envs > 1 else 50):                   # assemble a little, warm up
                a = env.step(a)[4].astype(np.float32)
            t0 = time.perf_counter()
            for _ in range(steps):
                a = env.step(a)[4].astype(np.float32)                  # the trainer's loop shape: host action from host outputs
            total_ms = (time.perf_counter() - t0) / steps * 1e3
            b = env._backend()
            act = torch.zeros((n_envs, n_a, 2), device=b.device)
            b.timer_start()
            for _ in range(steps):
                b.step(act)
            dev_ms = b.timer_stop() / steps
            rec = {"workload": label, "ms_per_step": total_ms, "agent_steps_per_s": n_envs * n_a / (total_ms * 1e-3),
                   "device_step_kernel_ms": dev_ms, "host_boundary_ms": total_ms - dev_ms,
                   "what": "one C call per step: pinned action staging -> H2D -> k_env (f64 rows) -> k_export (widen + transpose on "
                           "the device) -> ONE D2H into a pinned slot -> sync; host_boundary_ms = everything but the step kernel "
                           "(incl. the caller's astype of the next action)"}
            if n_envs == 1:
                rec["reference_cpu_ms_per_step"] = lat.get("reference_cpu_n30_ms_per_step")
            else:
                blk = b.obs_dim * n_envs * n_a * 8 + 3 * n_envs * n_a * 8 + n_envs * n_a
                rec["host_block_bytes"] = blk
                rec["d2h_GBps_if_all_boundary_time_were_the_copy"] = blk / ((total_ms - dev_ms) * 1e-3) / 1e9
            res.append(rec)
            env.close()
            del env
        except Exception as ex:
            res.append({"workload": label, "error": repr(ex)[:300]})
        torch.cuda.empty_cache()
    # (3) BASELINE config 0: the reference's own step (numpy glue + five native calls) at N = 8, native calls served by the
    #     reference's library on one host core vs by this library's legacy symbols (synchronous H2D / kernel / D2H per call)
    res.append({"workload": "assembly env, 8 agents x 1 env through the reference's five extern-C symbols (BASELINE config 0)",
                "what": "ms per reference-style step (numpy glue + 5 native calls): reference libAssemblyEnv.so on one core vs "
                        "this library's legacy symbols (GPU-backed; plumbing, not a fast path)", **lat})
    # (4) device-resident rollout step: env + fused policy (+ exploration noise) + replay push
    try:
        res.append(rollout_numbers(torch, args, device))
    except Exception as ex:
        res.append({"workload": "device rollout", "error": repr(ex)[:300]})
    return res


def rollout_numbers(torch, args, device):
    from marl_llm_amd.batched import SwarmBatch
    from marl_llm_amd.rollout import ChainedReplay, FusedPolicy, PolicyMLP, rollout
    from marl_llm_amd.shapes import r_avoid_for, synthetic_shape_set
    shapes = synthetic_shape_set()
    n_a, E, steps = 64, 4096, 100
    ng_max = max(np.asarray(g).shape[0] for g in shapes["grid_coords"])
    out = {"workload": "device-resident rollout step, 64 agents x 4096 envs: fused bf16 MFMA actor + exploration noise + env step "
                       "(+ ChainedReplay push)", "what": "ms per rollout step, everything on the device (marl_llm_amd/rollout.py)"}
    for tag, odt in (("f32_rows", torch.float32), ("bf16_rows", torch.bfloat16)):
        sb = SwarmBatch(n_env=E, n_agents=n_a, n_cells_max=ng_max, r_avoid=r_avoid_for(n_a, shapes), obs_dtype=odt, device=device)
        sb.set_shapes(shapes)
        obs = sb.reset(seed=args.seed)
        pol = FusedPolicy(PolicyMLP(obs_dim=sb.obs_dim).to(sb.device), device=sb.device)
        rep = ChainedReplay(8, E * n_a, sb.obs_dim, 2, sb.device, obs_dtype=odt)
        state = {"obs": obs}

        def run(k, replay):
            # fused path: noise in the policy kernel's epilogue, transition written straight into the ring (two launches / step)
            state["obs"], _ = rollout(sb, pol, k, state["obs"], replay=replay, noise_scale=0.1, track_reward=False,
                                      seed=args.seed, step0=state.get("t", 0))
            state["t"] = state.get("t", 0) + k

        for replay, key in ((None, "env_policy_noise_ms"), (rep, "env_policy_noise_replay_ms")):
            run(60, replay); torch.cuda.synchronize()
            t0 = time.perf_counter(); run(steps, replay); torch.cuda.synchronize()
            out[f"{tag}_{key}"] = (time.perf_counter() - t0) / steps * 1e3
        out[f"{tag}_agent_steps_per_s_with_replay"] = E * n_a / (out[f"{tag}_env_policy_noise_replay_ms"] * 1e-3)
        if tag == "f32_rows":                                          # the actor kernel alone, both arithmetic modes
            x = state["obs"].reshape(E * n_a, -1)
            pol3 = FusedPolicy(pol.module, device=sb.device, precision="bf16x3")
            for nm, f in (("actor_bf16_ms", pol), ("actor_bf16x3_ms", pol3)):
                for _ in range(10):
                    f(x)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(steps):
                    f(x)
                torch.cuda.synchronize()
                out[nm] = (time.perf_counter() - t0) / steps * 1e3
            pol3.close()
        sb.close()
        del sb, pol, rep
        torch.cuda.empty_cache()
    return out


def claim_stdout():
    """Rank 0's stdout must carry exactly ONE line: the JSON.  Native libraries (Gloo's "[Gloo] Rank 0 is connected
    ..." banner, RCCL, the HIP runtime) print to file descriptor 1 behind Python's back, so fd 1 is pointed at stderr
    for the whole run and the JSON line is written to a private duplicate of the original stdout."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(keep, "w")


def valu_bound(kernel_us, n_env):
    """Second roofline entry: the bound that actually binds k_env -- vector-instruction issue.
    instructions per launch by class come from the committed, source-hash-matched PMC summary (SQ_INSTS_VALU*, separate
    --pmc passes); issue cost per class from profiles/r02/microbench_valu_*.txt (cycles per wave-instruction on one
    SIMD with >= 2 resident waves: plain 32-bit 2.5, 3-operand / shift / compare / bit-count class 3.8-4.2, fp64 add /
    mul / fma 4.4-4.9, conversions 4.1, transcendental 8.1).  frac = modelled issue cycles / (SIMDs x launch cycles at
    2.4 GHz): the share of the launch during which the vector pipes are the resource in use."""
    import glob
    d = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "final_pmc_summary.json"))):
        try:
            c = json.load(open(f))
        except Exception:
            continue
        if c.get("_source_sha256") == source_hash() and "SQ_INSTS_VALU" in c and c.get("_n_env", 4096) == n_env:
            d = c
    if d is None:
        return {"bound": "valu", "insts_per_launch": None, "frac": None,
                "note": "no PMC summary for this kernel source (tools/collect_profiles.sh regenerates it)"}
    g = lambda k: float(d[k]["mean"]) if k in d else 0.0
    total = g("SQ_INSTS_VALU")
    f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
    cvt, i64 = g("SQ_INSTS_VALU_CVT"), g("SQ_INSTS_VALU_INT64")
    trans = g("SQ_INSTS_VALU_TRANS_F32") + g("SQ_INSTS_VALU_TRANS_F64")
    typed = "SQ_INSTS_VALU_ADD_F64" in d
    rest = total - f64 - cvt - i64 - trans
    cycles = f64 * 4.6 + cvt * 4.1 + i64 * 4.4 + trans * 8.1 + rest * 3.3 if typed else total * 3.6
    simds, clock = 256 * 4, 2.4e9
    return {"bound": "valu", "insts_per_launch": total, "insts_per_env": total / n_env,
            "issue_cycles_per_launch": cycles, "frac": cycles / (simds * kernel_us * 1e-6 * clock),
            "mix": {"fp64": f64, "cvt": cvt, "int64": i64, "transcendental": trans, "other": rest} if typed else None,
            "note": "modelled issue cycles (per-class counts x measured issue cost) / (1024 SIMDs x launch time x 2.4 GHz)"}


def main():
    args = parse()
    if args.cpu_worker:
        sys.exit(cpu_worker_main())
    if args.latency_worker:
        sys.exit(latency_worker_main())
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr); sys.exit(2)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))               # this process stays GPU-free; the children are the ranks

    from marl_llm_amd import dist_util as du
    rank, local_rank, world = du.rank_world()
    if world != args.gpus:
        print(f"bench.py: WORLD_SIZE={world} but --gpus {args.gpus}: refusing to report a mislabelled run", file=sys.stderr)
        sys.exit(2)
    out_stream = claim_stdout()                        # from here on fd 1 is stderr; the JSON goes to out_stream

    if args.dry_spawn:                                 # plumbing rehearsal: no GPU anywhere
        import torch
        if args.dry_fail_rank == rank:
            print(f"bench.py: rank {rank} exits 3 on request (--dry-fail-rank)", file=sys.stderr)
            sys.exit(3)
        du.init(backend="gloo")
        du.barrier()
        top = du.max_over_ranks(float(rank))
        ranks = du.gather_to_rank0(torch.tensor([[rank, local_rank, world]], dtype=torch.int64))
        per_rank = du.gather_to_rank0(torch.tensor([100.0 + rank], dtype=torch.float64))
        if rank == 0:
            out_stream.write(json.dumps({"dry_spawn": True, "n_gpus": world, "max_rank": top, "ranks": ranks[:, 0].tolist(),
                                         "local_ranks": ranks[:, 1].tolist(),
                                         "per_rank_kernel_us": per_rank.tolist()}) + "\n")
            out_stream.flush()
        du.barrier()
        du.shutdown()
        return

    # CPU-baseline workers start now, before this process initialises the GPU (they idle on stdin until the state exists)
    pool_all = pool_one = None
    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    if want_cpu:
        pool_one = start_cpu_workers(1)
        pool_all = start_cpu_workers(args.cpu_workers or host_cores())

    import torch
    if not torch.cuda.is_available():
        for p in (pool_one or []) + (pool_all or []):
            p.stdin.write("\n"); p.stdin.close()
        print("bench.py: no HIP device visible; the env step has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    du.init(backend="gloo" if args.rehearse_one_gpu else "nccl", local_rank=local_rank)
    barrier = du.barrier if world > 1 else None

    n_a, E = args.agents, args.envs
    # weak scaling: every rank owns its own slice [rank*E, (rank+1)*E) of the global env range
    sb, sy, r_avoid, timed, prewarm = measure(torch, n_a, E, args.state, args.steps, args.warmup, args.assemble_steps,
                                              args.seed, rank * E, f"cuda:{local_rank}", barrier)
    cpu_state = None
    if want_cpu:                                       # the state the timed GPU run starts from, for the CPU legs below
        cpu_state = capture_cpu_state(sb, sy)
    # The timed region runs FIRST, straight after >= 50 ms of untimed steps of the same loop (steady-state clocks): the
    # CPU baseline keeps the host busy for ~16 s, after which a 3 ms timed region would start from idle GPU clocks.
    n_prewarm = prewarm(args.prewarm_ms)
    dt, kernel_ms = timed()
    dt = du.max_over_ranks(dt, device="cpu" if (args.rehearse_one_gpu or world == 1) else sb.device)
    # (RCCL gathers device tensors, gloo host tensors)
    coll_dev = "cpu" if (args.rehearse_one_gpu or world == 1) else sb.device
    try:
        per_rank_us = du.gather_to_rank0(torch.tensor([kernel_ms * 1e3 / args.steps], dtype=torch.float64, device=coll_dev))
        per_rank_us = per_rank_us.cpu() if per_rank_us is not None else None
    except Exception as ex:                            # an optional key must never cost the bench line
        print(f"bench.py: per-rank gather failed: {ex!r}", file=sys.stderr)
        per_rank_us = None
    in_shape = float(sb.indices(False, False)["in_flags"].float().mean().item())
    alg64 = sb.algorithmic_bytes_per_step()
    sb.close()
    cpu = None
    if want_cpu:
        cpu = cpu_baseline(cpu_state, r_avoid, n_a, pool_all, pool_one, args.cpu_seconds)

    others = []
    if rank == 0 and world == 1 and not args.no_other_configs:
        from marl_llm_amd.shape_images import unpack_cells_npz
        fig = unpack_cells_npz(os.path.join(ROOT, "tests", "golden", "fig_cells.npz"))
        for (oa, oe, ost, osh, label) in (
                (32, 1024, "assembled", None, "BASELINE config 1"),
                (64, 4096, "scatter", None, "headline shape, reset() state distribution, U(-1,1) actions"),
                (64, 4096, "assembled", fig, "headline shape on the reference's own fig/*.png target shapes (38-40 lattice "
                                             "columns: 64-bit row masks instead of the synthetic set's 32-bit ones)"),
                (256, 4096, "assembled", None, "BASELINE config 4 (dense O(N^2) neighbour path)"),
                (64, 32768, "assembled", None, "BASELINE config 3's 8-GPU total on ONE GPU")):
            osb, osy, _, otimed, oprewarm = measure(torch, oa, oe, ost, 50, 10, 100, args.seed, 0, f"cuda:{local_rank}", shapes=osh)
            oprewarm(args.prewarm_ms)
            odt, okms = otimed()
            us = okms * 1e3 / 50
            b = survey_bytes(oa, osy["n_g"])
            others.append({"workload": f"assembly env, {oa} agents x {oe} envs, {ost} state"
                                       + (", reference fig shapes" if osh is not None else ""), "what": label,
                           "kernel_us": us, "agent_steps_per_s": oa * oe * 50 / odt,
                           "algorithmic_bytes_per_launch": b, "frac": b / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})
            osb.close()
            del osb, osy
            torch.cuda.empty_cache()
        others.extend(extra_configs(torch, args, f"cuda:{local_rank}"))

    if rank == 0:
        total_agent_steps = float(world) * E * n_a * args.steps
        value = total_agent_steps / dt
        launch_s = kernel_ms * 1e-3 / args.steps
        alg = survey_bytes(n_a, sy["n_g"])
        achieved = alg / launch_s / 1e9
        what = "assembled state, prior-policy actions" if args.state == "assembled" else "scatter state, U(-1,1) actions"
        workload = f"assembly env, {n_a} agents x {E} envs per GPU ({n_a} x {E * world} total), {what}"
        out = {
            "metric": "agent-steps/sec", "value": value, "unit": "agent-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (state, every index / flag / reward decision) + f32 (obs, prior, action I/O, pre-filters)",
            "data": "synthetic", "prewarm_steps": n_prewarm,
            "per_rank_kernel_us": per_rank_us.tolist() if per_rank_us is not None else None,
            "config": {"workload": workload,
                       "agents": n_a, "envs_per_gpu": E, "envs_total": E * world, "obs_dtype": "f32",
                       "state_dtype": "f64", "in_shape_fraction": round(in_shape, 3), "seed": args.seed,
                       "parallelism": f"env-sharded x{world}, no collective on the step path"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(f"assembly env, {n_a} agents x {E} envs per GPU, {args.state} state"),
                         "kernel": "k_env<%d,float,true>" % max(8, 1 << (n_a - 1).bit_length()), "kernel_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": alg,
                         "algorithmic_bytes_note": "SURVEY 8(d): 821 B per agent-step + 8 B per target cell per env",
                         "bytes_per_launch_this_build_dtypes": alg64,
                         "kernel_source_sha256": source_hash(),
                         "secondary": valu_bound(launch_s * 1e6, E)},
        }
        if cpu is not None:
            cpu["gpu_over_cpu"] = value / cpu["value"]
            cpu["one_core"]["gpu_over_cpu"] = value / cpu["one_core"]["value"]
            out["cpu_baseline"] = cpu
        if others:
            out["other_configs"] = others
        out_stream.write(json.dumps(out) + "\n")
        out_stream.flush()
    du.barrier()
    du.shutdown()


if __name__ == "__main__":
    main()
