"""The N > 1 path of bench.py on the GPU box.

* two ranks launched exactly as the driver launches them (python -m torch.distributed.run), sharing the box's one GPU and reducing through
  gloo (DYGNN_BENCH_BACKEND=gloo; on an 8-GPU node the backend is RCCL and every rank has its own device): process-group set-up, the graph
  generated once per node and mapped by the other rank, the round-robin batch deal, the ONE metric all-reduce after the timed region, the
  barrier + MAX-reduce of the timed region, the LastFM-shaped workload (BASELINE config 4) sharded over the ranks as `secondary.lastfm`,
  rank 0 printing the one JSON line with the whole-job value;
* ONE rank on the real `nccl` backend (= RCCL; DYGNN_BENCH_FORCE_DIST=1): init_process_group(device_id=...), barrier, the float64 SUM
  all-reduce and the MAX-reduce on device tensors — RCCL's first contact, in a fresh child process, so that the driver's 8-GPU run cannot
  fail for a reason a one-GPU box could have shown."""
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
    assert 0.0 <= d["mean_auc"] <= 1.0 and "cpu_baseline" not in d and "stages" not in d
    assert d["metric_allreduce"]["steps_counted"] == 2 * 6                                    # every rank's step counts arrived in the one all-reduce
    lf = d["secondary"]["lastfm"]                                                             # BASELINE config 4 at N ranks
    assert lf["n_gpus"] == 2 and lf["value"] > 0 and lf["roofline"]["kernel"] == "k_dygformer_fused3<8>" and 0.0 <= lf["mean_auc"] <= 1.0
    assert lf["metric_allreduce"]["steps_counted"] == 2 * lf["steps"]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("dygnn_bench_")]          # the shared graph files are gone


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_one_rank_on_rccl_runs_the_collectives_of_the_sharded_bench():
    env = dict(os.environ, DYGNN_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("DYGNN_BENCH_BACKEND", None)                                                      # the default: "nccl" = RCCL
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "2",
           "--workload", "tiny", "--prime-launches", "1", "--cpu-seconds", "0", "--secondary", "none"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["metric_allreduce"]["backend"] == "RCCL (nccl)" and d["metric_allreduce"]["steps_counted"] == 6
    assert 0.0 <= d["mean_auc"] <= 1.0
