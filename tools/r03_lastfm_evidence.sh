#!/bin/bash
# Round 3: counters + in-kernel phase shares of the config-4 kernel k_dygformer_fused3<8> (LastFM shape, L=512, P=8).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/lastfm
PHASE_WORKLOAD=lastfm PHASE_GROUPS=8 DYGNN_LIB_VARIANT=stamps python3 tools/phase_profile.py > gpurun_out/lastfm/phase.txt 2>&1 || echo "phase failed"
PHASE_WORKLOAD=wikipedia PHASE_GROUPS=20 DYGNN_LIB_VARIANT=stamps python3 tools/phase_profile.py > gpurun_out/lastfm/phase_wiki.txt 2>&1 || echo "phase wiki failed"
BENCH_ARGS="--workload lastfm --steps 16 --warmup 8 --fuse-steps 8" bash tools/pmc_profile.sh > gpurun_out/lastfm/pmc.log 2>&1
cp gpurun_out/pmc/summary.txt gpurun_out/lastfm/pmc_summary.txt
