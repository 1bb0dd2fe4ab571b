"""World-size-2 rehearsal of the multi-GPU path on CPU (gloo): env-range sharding, counter-based synthetic
inputs (a rank's slice equals the same slice of a single-process generation), barrier / max-over-ranks timing
and the single host-side gather.  The step itself has no collective, so there is nothing else to rehearse."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from marl_llm_amd import dist_util as du
from marl_llm_amd.shapes import synthetic_shape_set
from marl_llm_amd.synth import synthetic_batch
rank, local_rank, world = du.init(backend="gloo")
assert world == 2
E_total, N = 10, 8
b, e = du.shard_range(E_total, rank, world)
sy = synthetic_batch(e - b, N, synthetic_shape_set(), seed=226, env_offset=b)
full = synthetic_batch(E_total, N, synthetic_shape_set(), seed=226)
for k in ("cells", "n_g", "l_cell", "p", "dp"):
    assert np.array_equal(sy[k], full[k][b:e]), k
du.barrier()
t = du.max_over_ranks(1.0 + rank)
assert t == 2.0
g = du.gather_to_rank0(torch.from_numpy(sy["p"]))
if rank == 0:
    assert np.array_equal(g.numpy(), full["p"])
    print("GATHER_OK", tuple(g.shape))
else:
    assert g is None
du.barrier()
torch.distributed.destroy_process_group()
'''


def test_two_rank_sharding_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "GATHER_OK (10, 2, 8)" in out.stdout


def test_shard_range_covers_everything():
    from marl_llm_amd.dist_util import shard_range
    for n in (1, 7, 8, 4096, 32768):
        for world in (1, 2, 3, 8):
            cuts = [shard_range(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[k][1] == cuts[k + 1][0] for k in range(world - 1))
            sizes = [b - a for a, b in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun) must itself produce two ranks: --dry-spawn runs the same spawn /
    dist_util init / barrier / max-over-ranks / gather plumbing over gloo and exits before any GPU use."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-spawn"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    # rank 0's stdout is EXACTLY one line, the JSON: Gloo / RCCL banners and the other ranks' output go to stderr
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == [0, 1] and d["local_ranks"] == [0, 1] and d["max_rank"] == 1.0
    assert d["per_rank_kernel_us"] == [100.0, 101.0]          # one entry per rank, gathered through dist_util (rank order)


def test_bench_fails_fast_when_a_rank_dies():
    """A rank that exits early must stop the whole run at once with its exit code, not leave rank 0 waiting in a
    rendezvous until a timeout (bench.py spawn_ranks polls all children)."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-spawn", "--dry-fail-rank", "1"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 3, (out.returncode, out.stderr[-1000:])
    assert "rank 1 failed (exit code 3)" in out.stderr and out.stdout.strip() == ""
    assert time.time() - t0 < 120


def test_bench_refuses_a_world_size_mismatch():
    """Under a launcher that gives WORLD_SIZE != --gpus the run must fail loudly instead of printing n_gpus of its own."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29613")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-spawn"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "WORLD_SIZE=1 but --gpus 2" in out.stderr
