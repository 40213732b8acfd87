#!/bin/bash
# Reproduces the committed profiles of a round on the GPU box (run through gpurun):
#   tools/profile_round.sh r01
# 1. bench.py JSON line                       -> gpurun_out/<tag>_bench_final13682.json
# 2. rocprofv3 --kernel-trace --stats         -> gpurun_out/<tag>_final13682_kernel_stats.csv
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel-trace only)
#                                             -> gpurun_out/<tag>_final13682_pmc_summary.json
# Copy the three files into profiles/ afterwards (gpurun_out/ is scratch).
set -u
TAG=${1:-r02}
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
timeout 900 python3 bench.py --steps 10 --warmup 3 > $OUT/${TAG}_bench_final13682.json 2> $OUT/${TAG}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
cp $OUT/prof_stats/st_kernel_stats.csv $OUT/${TAG}_final13682_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/prof_fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/prof_write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/prof_write.log 2>&1
python3 tools/pmc_summary.py $OUT/prof_fetch/f_counter_collection.csv $OUT/prof_write/w_counter_collection.csv > $OUT/${TAG}_final13682_pmc_summary.json
head -c 600 $OUT/${TAG}_bench_final13682.json; echo
head -12 $OUT/${TAG}_final13682_kernel_stats.csv
