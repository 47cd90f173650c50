#!/bin/bash
# HBM-side traffic of the TGAT attention kernel inside the TGAT bench (run on the GPU box via gpurun): separate rocprofv3 --pmc passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), --kernel-trace only.  Summary: tools/pmc_summary.py gpurun_out/pmc_tgat_attn k_tgat_attn_pair
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_tgat_attn
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "rdsz TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/bench_tgat.py" --steps 32 --warmup 1 --cpu-seconds 0 > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
done
python3 "$ROOT/tools/pmc_summary.py" "$OUT" k_tgat_attn_pair
