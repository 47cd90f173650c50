#!/bin/bash
# Instruction mix of the fused DyGFormer kernel (round 3): how many VALU / MFMA / LDS / SALU / VMEM instructions a launch issues and how many
# cycles the VALU and the matrix pipe are busy — the accounting behind DESIGN.md §4.3 "Round 3" (a VALU instruction takes matrix-pipe time).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_mix
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --cpu-seconds 0 --secondary none ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
run a SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
run c SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_WAVES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES
python3 "$ROOT/tools/pmc_summary.py" "$OUT" | tee "$OUT/summary.txt"
