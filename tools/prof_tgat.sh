#!/bin/bash
# rocprofv3 --kernel-trace --stats of the fused `recent` TGAT calls (tools/bench_tgat.py --plain) -> gpurun_out/r03_tgat_kernel_stats.csv
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_tgat3 -- python3 $ROOT/tools/bench_tgat.py --plain > $ROOT/gpurun_out/prof_tgat3.json 2> $ROOT/gpurun_out/prof_tgat3.err
find $ROOT/gpurun_out/prof_tgat3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $ROOT/gpurun_out/r03_tgat_kernel_stats.csv
rm -rf $ROOT/gpurun_out/prof_tgat3
python3 - <<P
import csv
rows=list(csv.DictReader(open("$ROOT/gpurun_out/r03_tgat_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU time total: {tot/1e6:.3f} ms")
for r in rows[:16]:
    print(f"{r['Name'][:80]:80s} {r['Calls']:>5s} {float(r['TotalDurationNs'])/1e6:8.3f} ms {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:6.2f}%")
P
