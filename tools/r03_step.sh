#!/bin/bash
# one GPU round trip of the builder: the GPU tests, the driver's bench command, the in-kernel phase shares of both DyGFormer shapes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/step
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/step/tests.log 2>&1 || { tail -30 gpurun_out/step/tests.log; exit 1; }
tail -1 gpurun_out/step/tests.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 ${BENCH_ARGS:-} > gpurun_out/step/bench.json 2> gpurun_out/step/bench.err || { tail -20 gpurun_out/step/bench.err; exit 1; }
PHASE_WORKLOAD=lastfm PHASE_GROUPS=8 DYGNN_LIB_VARIANT=stamps timeout -k 10 200 python3 tools/phase_profile.py > gpurun_out/step/phase_lastfm.txt 2>&1 || echo "phase failed"
PHASE_WORKLOAD=wikipedia PHASE_GROUPS=20 DYGNN_LIB_VARIANT=stamps timeout -k 10 200 python3 tools/phase_profile.py > gpurun_out/step/phase_wiki.txt 2>&1 || echo "phase wiki failed"
python3 - <<'P'
import json
d=json.load(open("gpurun_out/step/bench.json"))
print("headline", d["value"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"], "per_call", d.get("stages",{}).get("per_call",{}).get("value"))
for k,v in d.get("secondary",{}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("parity",{}).get("ok"), v.get("error"))
print("cpu", d.get("cpu_baseline"), "parity", d.get("parity"))
print("full_span", d.get("stages",{}).get("full_span"))
print("wall", d.get("wall_s"))
P
