#!/bin/bash
# the dataflow sweeps' inline chain product by level width: tools/sweep_tri_inline.sh  (ms per LM iteration, sweeps ms)
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
for w in final-13682 ladybug-1723 venice-1778 synthetic-10k; do
  for t in 0 4 8 16 32 1000; do
    python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-other-variants --no-other-workloads --opt tri_inline=$t 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); print('$w tri_inline=$t', round(d['value'],3), 'sweeps', round(d['stages_ms_per_step']['tri_solve'],3))"
  done
done
