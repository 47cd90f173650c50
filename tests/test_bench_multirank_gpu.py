"""The N > 1 path of bench.py on the GPU box: two ranks launched exactly as the driver launches them (python -m torch.distributed.run),
sharing the box's one GPU and reducing through gloo (DYGNN_BENCH_BACKEND=gloo; on an 8-GPU node the backend is RCCL and every rank has
its own device).  Covers what no CPU test can: process-group set-up, the round-robin batch deal, the per-launch metric all-reduce on
device tensors, the barrier + MAX-reduce of the timed region, rank 0 printing the one JSON line with the whole-job value."""
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


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_print_one_whole_job_line():
    env = dict(os.environ, DYGNN_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--workload", "tiny", "--prime-launches", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["scaling"] == "weak" and d["unit"] == "edges/s"
    assert d["value"] > 0 and abs(d["value"] - 2 * 6 * 200 / (d["ms_per_step"] * 6 * 1e-3)) <= 1e-3 * d["value"]     # whole-job aggregate
    assert 0.0 <= d["mean_auc"] <= 1.0 and "secondary" not in d and "cpu_baseline" not in d                        # N > 1: the contract fields only
