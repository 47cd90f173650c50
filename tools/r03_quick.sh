#!/bin/bash
# quick GPU check of the fused DyGFormer kernel: its parity tests, both shapes timed (no CPU legs), in-kernel phase shares
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/quick
timeout -k 10 600 python3 -m pytest tests/test_dygformer_gpu.py tests/test_train_gpu.py tests/test_gradients_golden.py -m gpu -x -q > gpurun_out/quick/tests.log 2>&1 || { tail -30 gpurun_out/quick/tests.log; exit 1; }
tail -1 gpurun_out/quick/tests.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary lastfm > gpurun_out/quick/bench.json 2> gpurun_out/quick/bench.err || { tail -20 gpurun_out/quick/bench.err; exit 1; }
PHASE_WORKLOAD=lastfm PHASE_GROUPS=8 DYGNN_LIB_VARIANT=stamps timeout -k 10 200 python3 tools/phase_profile.py > gpurun_out/quick/phase_lastfm.txt 2>&1 || echo "phase failed"
PHASE_WORKLOAD=wikipedia PHASE_GROUPS=20 DYGNN_LIB_VARIANT=stamps timeout -k 10 200 python3 tools/phase_profile.py > gpurun_out/quick/phase_wiki.txt 2>&1 || echo "phase wiki failed"
python3 - <<'P'
import json
d=json.load(open("gpurun_out/quick/bench.json"))
print("headline", d["value"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"], "per_call", d.get("stages",{}).get("per_call",{}).get("value"))
for k,v in d.get("secondary",{}).items():
    print(k, v.get("value"), v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("error"))
P
grep "proj \|total\|windows" gpurun_out/quick/phase_lastfm.txt gpurun_out/quick/phase_wiki.txt
