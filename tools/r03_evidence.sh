#!/bin/bash
# Round-3 evidence run: counter passes of the fused DyGFormer kernel on both shapes (-> profiles/r03_*_pmc_summary.txt, *_traffic.json),
# of the neighbour-lookup kernels, and the kernel-trace statistics of the driver's bench command.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
E=gpurun_out/evidence; mkdir -p $E
PMC_OUT=$ROOT/$E/pmc_wiki bash tools/pmc_profile.sh > $E/pmc_wiki.log 2>&1
python3 tools/pmc_traffic.py $E/pmc_wiki wikipedia 2000 4000 180600 $E/r03_wiki_traffic.json > $E/traffic_wiki.txt 2>&1
cp $E/pmc_wiki/summary.txt $E/r03_wiki_pmc_summary.txt
BENCH_ARGS="--workload lastfm --steps 16 --warmup 8 --fuse-steps 8" PMC_OUT=$ROOT/$E/pmc_lastfm bash tools/pmc_profile.sh > $E/pmc_lastfm.log 2>&1
python3 tools/pmc_traffic.py $E/pmc_lastfm lastfm 3200 3200 1430000 $E/r03_lastfm_traffic.json > $E/traffic_lastfm.txt 2>&1
cp $E/pmc_lastfm/summary.txt $E/r03_lastfm_pmc_summary.txt
bash tools/pmc_sampler.sh > $E/pmc_sampler.log 2>&1; cp gpurun_out/pmc_sampler/summary.txt $E/r03_sampler_pmc_summary.txt
python3 tools/bench_sampler.py --queries 2000000 --reps 5 > $E/r03_sampler_stage_bench.jsonl 2> $E/sampler.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$E/stats -- python3 $ROOT/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary none > $ROOT/$E/stats_bench.json 2> $ROOT/$E/stats.err )
find $E/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $E/r03_kernel_stats.csv
find $E/stats -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $E/r03_kernel_trace.csv
rm -rf $E/stats $E/pmc_wiki/*/ $E/pmc_lastfm/*/ gpurun_out/pmc_sampler/*/
cat $E/traffic_wiki.txt $E/traffic_lastfm.txt; head -5 $E/r03_kernel_stats.csv
