#!/bin/bash
# HBM traffic counters of the neighbour-lookup kernels (k_sample_recent, k_find_before, k_window_fill, k_cooccurrence) under
# tools/bench_sampler.py: separate rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), --kernel-trace only.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_sampler
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "rdsz TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/bench_sampler.py" --queries 2000000 --reps 3 > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
done
for k in k_sample_recent k_find_before k_window_fill k_cooccurrence; do echo "== $k"; python3 "$ROOT/tools/pmc_summary.py" "$OUT" $k; done | tee "$OUT/summary.txt"
