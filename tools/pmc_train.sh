#!/bin/bash
# Hardware-counter passes over the training step's kernels (run on the GPU box via gpurun): one rocprofv3 run per counter group,
# --kernel-trace only; per-kernel means under gpurun_out/pmc_train/summary.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_train
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/bench_train.py" --steps 6 --warmup 2 --cpu-seconds 0 > "$OUT/$name.log" 2>&1 || echo "pass $name failed"
}
run sq    SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32
run fetch FETCH_SIZE
run write WRITE_SIZE
run misc  GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU
{
for k in "k_dygformer_fused3" "k_ffn_bwd" "k_attn_bwd" "k_dw_grouped"; do
  python3 "$ROOT/tools/pmc_summary.py" "$OUT" "$k"
done
} | tee "$OUT/summary.txt"
