#!/bin/bash
# end-of-round run: GPU tests, smoke(), the driver's bench command (-> r03_bench_end_of_round.json), bench.py with no flags, the LastFM workload stand-alone
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
O=gpurun_out/final; mkdir -p $O
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
timeout -k 10 200 python3 __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_end_of_round.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
timeout -k 10 300 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err || { tail -20 $O/bench_default.err; exit 1; }
timeout -k 10 300 python3 bench.py --workload lastfm --steps 16 --warmup 8 --fuse-steps 8 --secondary none > $O/bench_lastfm.json 2> $O/bench_lastfm.err || { tail -20 $O/bench_lastfm.err; exit 1; }
python3 - <<'P'
import json
for f in ("r03_bench_end_of_round","bench_default","bench_lastfm"):
    d=json.load(open(f"gpurun_out/final/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"], d["config"]["steps_per_launch"], "wall", d.get("wall_s"), "parity", d.get("parity",{}).get("ok"))
    for k,v in d.get("secondary",{}).items():
        print("   ", k, v.get("value"), v.get("ms_per_step"), v.get("roofline",{}).get("frac"), v.get("parity",{}).get("ok"), v.get("error"))
P
