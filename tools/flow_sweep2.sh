# auto model vs manual thresholds.  usage: tools/flow_sweep2.sh <tag>
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
O=gpurun_out; T=${1:-r04}
run() {
  w=$1; shift; st=$1; shift
  python3 bench.py --workload $w --steps $st --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
try:
    b=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=b.get('stages_ms_per_step',{})
    print('%-16s %-44s %8.3f ms/it  factor %.3f  setup %s' % ('$w', '$*', b['value'], st.get('factor'), b.get('setup_wall_s', b.get('setup_s'))))
except Exception as e: print('$w $* FAILED', e)
"
}
for w in final-13682 ladybug-1723 venice-1778 sphere2500 synthetic-10k final-13682-hub; do
  run $w 10 --opt factor_flow=0
  run $w 10
  run $w 10 --opt factor_flow=8 --opt factor_flow_rows=32
  run $w 10 --opt factor_flow=16 --opt factor_flow_rows=48
done 2>&1 | tee $O/${T}_flow_sweep2.txt
