"""The RCCL branch on one GPU: a one-rank process group over backend "nccl" (= RCCL on ROCm) is legal, so the calls the
8-GPU launch makes first -- init_process_group with a device id, barrier, the MAX / SUM all-reduces of bench.py, and
Summarize's all-reduce (batch.go:140-158) -- can run, and be checked, before anyone spends a multi-GPU lease."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1",
                "HSA_ENABLE_IPC_MODE_LEGACY": "0", "PYTHONPATH": ROOT + os.pathsep + env.get("PYTHONPATH", "")})
    return env


SCRIPT = r"""
import json, sys
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from fennec_amd import batch as fb
res = [fb.BatchResult(Index=i, OriginalSize=1000 + 7 * i, CompressedSize=400 + 3 * i, SSIM=0.9 + 0.001 * i, Quality=80) for i in range(37)]
res.append(fb.BatchResult(Index=37, Err="boom", has_result=False))
loc = fb.summarize_local(res)
red = fb.summarize_distributed(res, device="cuda", force=True)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
out = {"backend": dist.get_backend(), "local": [loc.Total, loc.Succeeded, loc.Failed, loc.TotalSaved, loc.AvgSSIM],
       "reduced": [red.Total, red.Succeeded, red.Failed, red.TotalSaved, red.AvgSSIM], "max": float(t.item())}
dist.destroy_process_group()
print(json.dumps(out))
"""


@pytest.mark.gpu
def test_summarize_over_rccl_one_rank():
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=_env(), capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert lines, (p.stdout[-1500:], p.stderr[-1500:])
    out = json.loads(lines[-1])
    assert out["backend"] == "nccl"
    assert out["reduced"] == out["local"] and out["max"] == 1.25


@pytest.mark.gpu
def test_bench_line_with_the_process_group_up():
    """bench.py under torch.distributed.run with one rank: _dist_setup initialises RCCL, the timed region's barrier and
    reductions and the batch object's Summarize go through it, rank 0 prints the line."""
    env = _env()
    env["FENNEC_BENCH_FORCE_DIST"] = "1"
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", env["MASTER_PORT"], os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--no-extras", "--no-cpu-baseline", "--batch-items", "24", "--batch-files", "4", "--prewarm", "0.05"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["batch"]["summarize"]["Total"] == 24 and "RCCL" in line["batch"]["summarize"]["how"]
    # the `dist` object: who took part, all-gathered over RCCL itself
    d = line["dist"]
    assert d["backend"].startswith("rccl") and d["world"] == 1 and d["devices_distinct"] is True
    assert len(d["rccl_ranks_seen"]) == 1 and d["rccl_ranks_seen"][0]["rank"] == 0 and d["rccl_ranks_seen"][0]["device"] == 0
    assert d["rccl_ranks_seen"][0]["pci"] and "queue" in d
