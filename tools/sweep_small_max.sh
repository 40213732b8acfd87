export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
for o in "" "--opt panel_small_max=0" "--opt panel_small_max=16" "--opt panel_small_max=0 --opt update_small_max=0" "--opt update_small_max=16" ""; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-variants --no-other-workloads $o 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][0]); st=d['stages_ms_per_step']
print('[$o]', round(d['value'],3), 'factor', round(st['factor'],3), 'tri', round(st['tri_solve'],3))"
done
