#!/bin/bash
# A/B of implementation switches on ONE box: tools/ab_bench.sh "<opts A>" "<opts B>" [reps]   (opts: "--opt a=1 --opt b=0" or "")
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
A=$1; B=$2; R=${3:-3}
for k in $(seq 1 $R); do
  for v in A B; do
    if [ $v = A ]; then O=$A; else O=$B; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-variants --no-other-workloads $O 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); st=d['stages_ms_per_step']
print('$v [$O]', round(d['value'],3), 'final_cost', repr(d.get('final_cost')), 'stage sum', round(sum(st.values()),3), {k:round(x,3) for k,x in st.items() if x>0})"
  done
done
