cd $GRAFT_REPO_ROOT
nproc; python3 -c "import os; print(len(os.sched_getaffinity(0)))"
for cfg in "256 close cores" "256 spread cores" "256 false -" "64 spread cores" "32 close cores" "1 close cores"; do
  set -- $cfg
  env="OMP_NUM_THREADS=$1 OMP_PROC_BIND=$2"
  [ "$3" != "-" ] && env="$env OMP_PLACES=$3"
  env $env python3 -c "
import sys, json; sys.path.insert(0, '.'); import bench
r=bench.cpu_baseline_worker('final-13682', 0.0243629, 'selfcal'); print('$cfg', round(r['value'],1), 'ms/iter', int(r['obs_per_s']), 'obs/s')" 2>&1 | tail -2
done
