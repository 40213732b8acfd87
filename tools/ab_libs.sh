#!/bin/bash
# A/B of several BUILDS of the library on one box: tools/ab_libs.sh <reps> <lib> [<lib> ...]   (paths relative to the repo root; APEXGPU_LIB)
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
R=$1; shift
for k in $(seq 1 $R); do
  for lib in "$@"; do
    APEXGPU_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-variants --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); st=d['stages_ms_per_step']
print('$lib', round(d['value'],3), 'final_cost', repr(d.get('final_cost')), {k:round(x,3) for k,x in st.items() if x>0})"
  done
done
