#!/bin/bash
# rocprofv3 kernel stats of the training step (run on the GPU box): gpurun_out/prof_tr/tr_kernel_stats.csv + a short summary on stdout
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT/gpurun_out/prof_tr"
export DYGNN_BENCH_TRAIN_PLAIN=1
rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/prof_tr" -o tr -- python3 "$ROOT/tools/bench_train.py" --steps 8 --warmup 2 --cpu-seconds 0 ${TRAIN_ARGS:-} > "$ROOT/gpurun_out/prof_tr.log" 2>&1
python3 - "$ROOT/gpurun_out/prof_tr/tr_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f"GPU time per step (10 steps): {tot / 1e7:.3f} ms")
for r in rows[:int(__import__('os').environ.get('TOPK', '16'))]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['TotalDurationNs']) / 1e7:8.4f} ms/step {float(r['AverageNs']) / 1e3:8.1f} us")
PY
