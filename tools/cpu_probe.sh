# What the GPU box's host side really offers (bench.py's cpu_baseline must size its thread counts from this, not from
# os.cpu_count()): hardware threads, affinity mask, cgroup CPU quota, NUMA layout; then the oracle's iteration time on the
# 333-camera sample for several thread counts and OpenMP wait policies.
cd $GRAFT_REPO_ROOT
echo "nproc $(nproc)  affinity $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))')"
echo "cgroup v2 cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "cgroup v1 quota/period: $(cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null) / $(cat /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null)"
lscpu | grep -E "^CPU\(s\)|Thread|Core|Socket|NUMA|Model name" 
free -g | head -2
make -C oracle native >/dev/null 2>&1
for cfg in "1 passive" "8 passive" "32 passive" "32 active" "64 passive" "128 passive" "256 passive" "256 active"; do
  set -- $cfg
  OMP_NUM_THREADS=$1 OMP_WAIT_POLICY=$2 OMP_PROC_BIND=false python3 -c "
import sys, json; sys.path.insert(0, '.'); import bench
r=bench.cpu_baseline_worker('final-13682', 0.0243629, 'selfcal'); print('threads $1 wait $2:', round(r['value'],1), 'ms/iter')" 2>&1 | tail -1
done
