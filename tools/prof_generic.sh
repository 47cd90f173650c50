#!/bin/bash
# rocprofv3 kernel stats of any bench tool: tools/prof_generic.sh <name> <python script> [args...] -> gpurun_out/prof_<name>/<name>_kernel_stats.csv
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof_$NAME"
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_$NAME" -o "$NAME" -- python3 "$ROOT/$1" "${@:2}" > "$ROOT/gpurun_out/prof_$NAME.log" 2>&1
python3 - "$ROOT/gpurun_out/prof_$NAME/${NAME}_kernel_stats.csv" <<'PY'
import csv, sys, os
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"GPU time total: {tot / 1e6:.3f} ms, launches {sum(int(r['Calls']) for r in rows)}")
for r in rows[:int(os.environ.get('TOPK', '16'))]:
    print(f"{r['Name'][:84]:84s} {r['Calls']:>6s} {float(r['TotalDurationNs']) / 1e6:9.3f} ms {float(r['AverageNs']) / 1e3:8.1f} us {r['Percentage']:>6s}%")
PY
