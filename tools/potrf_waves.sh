#!/bin/bash
for mode in 1 6 8; do
  echo "== potrf_lookahead=$mode"
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt potrf_lookahead=$mode 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); st = j['stages_ms_per_step']
print('final-13682 ms', round(j['value'], 2), 'factor', round(st['factor'], 2), 'final cost', j['final_cost'])"
  python bench.py --workload sphere2500 --steps 20 --warmup 3 --no-cpu-baseline --opt potrf_lookahead=$mode 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readlines()[-1]); st = j['stages_ms_per_step']
print('sphere2500 ms', round(j['value'], 3), 'factor', round(st['factor'], 3))"
done
