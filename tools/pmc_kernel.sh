#!/bin/bash
# usage: tools/pmc_kernel.sh <tag> <kernel-regex> "<counters>" [schur_bench args...]
# one rocprofv3 --pmc pass (kernel-trace only) over tools/schur_bench.py; prints the per-dispatch averages of the counters
tag=$1; regex=$2; ctrs=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
rm -rf $out
timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "$regex" -d $out -o p --output-format csv -- python3 tools/schur_bench.py "$@" > gpurun_out/pmc_${tag}.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()): print(f"   {c:28s} n={len(vals):3d} mean={sum(vals)/len(vals):.4g}")
PY
