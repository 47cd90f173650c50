#!/bin/bash
# Hardware counters of the general GEMM inside the TGAT bench (run on the GPU box via gpurun): one rocprofv3 pass per group.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_gemm
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/bench_tgat.py" --steps 32 --warmup 1 --cpu-steps 0 > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
  cp "$OUT/$name"/*/*counter_collection.csv "$OUT/$name.csv" 2>/dev/null; rm -rf "$OUT/$name"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE
