# tools/ab_other_workloads.sh <option> <v1> <v2>: A/B of an implementation switch on the other workloads
O=${1:-flood_gate}; A=${2:-50}; B=${3:-0}
for w in sphere2500 ladybug-1723 venice-1778 synthetic-10k; do for g in $A $B $A $B; do
timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --opt $O=$g 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); st=b['stages_ms_per_step']; print('$w $O=$g total', round(b['value'],3), 'factor', round(st.get('factor',0),3))"
done; done
for g in $A $B; do timeout 300 python bench.py --workload final-13682-hub --steps 4 --warmup 1 --no-cpu-baseline --opt $O=$g 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('hub $O=$g', round(b['value'],2))"; done
