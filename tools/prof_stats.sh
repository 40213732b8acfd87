#!/bin/bash
# usage: tools/prof_stats.sh <tag> [bench args...]  -> gpurun_out/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o run -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
head -12 $f | cut -c1-200
