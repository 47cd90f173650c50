#!/bin/bash
# Hardware-counter passes for the fused kernel (run on the GPU box via gpurun).  One rocprofv3 run per
# counter group, --kernel-trace only (no other trace domains), summaries under gpurun_out/pmc/.
# The profiled command is the driver's: bench.py --steps 20 --warmup 5 (CPU legs and secondary workloads off).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${PMC_OUT:-$ROOT/gpurun_out/pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --cpu-seconds 0 --secondary none ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
run sq    SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32
run fetch FETCH_SIZE
run write WRITE_SIZE
run rdsz  TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
run misc  GRBM_GUI_ACTIVE TA_BUSY_avr SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
python3 "$ROOT/tools/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"
