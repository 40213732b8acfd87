# kernel trace of a few LM iterations + the factorisation timeline.  usage: tools/timeline_run.sh <tag> [bench args]
TAG=${1:-r04}; shift
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
rm -rf $OUT/prof_tl
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_tl -o tl --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline "$@" > $OUT/prof_tl.log 2>&1
python3 tools/factor_timeline.py $OUT/prof_tl/tl_kernel_trace.csv > $OUT/${TAG}_factor_timeline.txt 2>&1
cat $OUT/${TAG}_factor_timeline.txt
