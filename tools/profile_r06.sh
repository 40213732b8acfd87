#!/bin/bash
# Round-6 evidence in one gpurun call (copy what is wanted from gpurun_out/ into profiles/ afterwards):
#   tools/profile_r06.sh [parts]     parts: any of  bench stats pmc timeline other lockstep pairs micro  (default: all)
set -u
PARTS=${1:-"bench stats pmc timeline other lockstep pairs micro"}
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out; T=r06
has() { [[ " $PARTS " == *" $1 "* ]]; }
if has bench; then
  timeout 900 python3 bench.py --steps 20 --warmup 5 > $O/${T}_bench_final13682.json 2> $O/${T}_bench.err
  python3 -c "import json; b=json.load(open('$O/${T}_bench_final13682.json')); print('final-13682', b['value'], b['stages_ms_per_step'], 'iterative', b.get('iterative_ms'), 'implicit', b.get('fallback_implicit'), b['roofline']['frac'], b['roofline'].get('fp64_pipe_frac'), b['factor']['frac'], b['setup_s'], b['cpu_baseline'].get('value'))"
fi
if has stats; then
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o st --output-format csv -- python3 bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-other-variants --no-other-workloads > $O/prof_stats.log 2>&1
  cp $O/prof_stats/st_kernel_stats.csv $O/${T}_final13682_kernel_stats.csv; head -14 $O/${T}_final13682_kernel_stats.csv | cut -c1-160
fi
if has pmc; then
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fetch -o f --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-variants --no-other-workloads > $O/prof_fetch.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_write -o w --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-variants --no-other-workloads > $O/prof_write.log 2>&1
  python3 tools/pmc_summary.py $O/prof_fetch/f_counter_collection.csv $O/prof_write/w_counter_collection.csv > $O/${T}_final13682_pmc_summary.json; head -c 700 $O/${T}_final13682_pmc_summary.json; echo
fi
if has timeline; then
  tools/timeline_run.sh ${T} --no-other-variants --no-other-workloads > /dev/null 2>&1; head -8 $O/${T}_factor_timeline.txt
  tools/timeline_run.sh ${T}_ladybug --workload ladybug-1723 > /dev/null 2>&1; head -5 $O/${T}_ladybug_factor_timeline.txt
fi
if has other; then
  for w in ladybug-1723 venice-1778 sphere2500; do timeout 900 python3 bench.py --workload $w --steps 10 --warmup 3 > $O/${T}_bench_${w//-/}.json 2>/dev/null; done
  timeout 600 python3 bench.py --workload final-13682-hub --steps 5 --warmup 2 --no-cpu-baseline --no-other-variants --no-other-workloads > $O/${T}_bench_final13682hub.json 2>/dev/null
  timeout 600 python3 bench.py --mode ba --steps 10 --warmup 3 --no-cpu-baseline > $O/${T}_bench_final13682_ba6.json 2>/dev/null
  timeout 600 python3 bench.py --workload synthetic-10k --steps 10 --warmup 3 --no-cpu-baseline > $O/${T}_bench_synthetic10k.json 2>/dev/null
  timeout 600 python3 bench.py --workload synthetic-10k --variant implicit --steps 3 --warmup 1 --no-cpu-baseline > $O/${T}_bench_synthetic10k_implicit.json 2>/dev/null
  for f in ladybug1723 venice1778 sphere2500 final13682hub final13682_ba6 synthetic10k synthetic10k_implicit; do python3 - $f <<'PY'
import json, sys
try:
    b = json.load(open(f"gpurun_out/r06_bench_{sys.argv[1]}.json")); cb = b.get("cpu_baseline") or {}
    print(sys.argv[1], round(b["value"], 3), b["unit"], "factor", round(b["stages_ms_per_step"].get("factor", 0), 3), "| cpu", cb.get("value"), cb.get("solve"), cb.get("cores"),
          "dense", (cb.get("dense_solve") or {}).get("value"), "| iterative", b.get("iterative_ms"), "implicit", b.get("fallback_implicit"), "| pcg", (b.get("pcg_iterations_per_step") or [])[:3])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
  done
fi
if has lockstep; then timeout 1500 python3 tools/dist_lockstep_times.py final-13682 2,4,8 > $O/${T}_lockstep_rank_times.txt 2>&1; cat $O/${T}_lockstep_rank_times.txt | grep -v "^  rank [1-6]"; fi
if has pairs; then   # counters of the pair kernel in the queued layout (form 4, the default)
  timeout 900 tools/profile_pairs.sh ${T}_queued > /dev/null 2>&1; mv $O/${T}_queued_k_schur_pairs_counters.txt $O/${T}_pairs_queued_counters.txt
  grep -E "SQ_LDS_BANK|TCC_MISS|SQ_INSTS_VALU |SQ_INSTS_LDS|SQ_INSTS_SALU|^form" $O/${T}_pairs_queued_counters.txt
  python3 tools/schur_bench.py --forms 4,3 --iters 10 2>&1 | grep -E "^form" > $O/${T}_pairs_forms.txt; cat $O/${T}_pairs_forms.txt
fi
if has micro; then
  tools/potrf_bench > $O/${T}_potrf_bench.txt 2>&1; grep -A4 "mode 12,  1" $O/${T}_potrf_bench.txt | grep -v stamps
  tools/lat_bench > $O/${T}_lat_bench.txt 2>&1; cat $O/${T}_lat_bench.txt
  tools/flow_bench 18 5 > $O/${T}_flow_bench_T18.txt 2>&1; head -8 $O/${T}_flow_bench_T18.txt
fi
