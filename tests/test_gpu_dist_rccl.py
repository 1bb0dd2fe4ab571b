"""The collectives bench.py issues at N > 1 (barrier, max-over-ranks of the wall time, per-rank kernel-time gather), run
through RCCL itself: a one-rank "nccl" process group on the box's GPU, device tensors as RCCL needs them.  (Two ranks on
one card are refused by RCCL -- "duplicate GPU" -- so the two-rank rehearsals use gloo: tests/test_dist_gloo.py.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
import torch.distributed as dist
from marl_llm_amd import dist_util as du
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
du.barrier()
dev = torch.device("cuda", 0)
assert du.max_over_ranks(3.5, device=dev) == 3.5
g = du.gather_to_rank0(torch.tensor([87.5], dtype=torch.float64, device=dev))
assert g.device.type == "cuda" and g.cpu().tolist() == [87.5]
du.barrier()
du.shutdown()
print("RCCL_OK")
'''


@pytest.mark.gpu
def test_bench_collectives_through_rccl(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script), ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "RCCL_OK" in out.stdout
