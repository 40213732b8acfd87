for w in sphere2500 ladybug-1723 venice-1778 synthetic-10k; do for g in 50 0 50 0; do
timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --opt flood_gate=$g 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); st=b['stages_ms_per_step']; print('$w gate=$g total', round(b['value'],3), 'factor', round(st.get('factor',0),3))"
done; done
timeout 300 python bench.py --workload final-13682-hub --steps 4 --warmup 1 --no-cpu-baseline --opt flood_gate=50 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('hub gate=50', round(b['value'],2))"
timeout 300 python bench.py --workload final-13682-hub --steps 4 --warmup 1 --no-cpu-baseline --opt flood_gate=0 2>/dev/null | python3 -c "
import json,sys; b=json.loads(sys.stdin.read()); print('hub gate=0', round(b['value'],2))"
