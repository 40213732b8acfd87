export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
O=gpurun_out
timeout 600 python3 bench.py --workload ladybug-1723 --steps 10 --warmup 3 > $O/r03_bench_ladybug1723.json 2>/dev/null
timeout 600 python3 bench.py --workload venice-1778 --steps 10 --warmup 3 > $O/r03_bench_venice1778.json 2>/dev/null
timeout 600 python3 bench.py --workload sphere2500 --steps 10 --warmup 3 > $O/r03_bench_sphere2500.json 2>/dev/null
timeout 600 python3 bench.py --workload final-13682-hub --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_final13682_hub.json 2>/dev/null
timeout 600 python3 bench.py --variant iterative --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_final13682_iterative.json 2>/dev/null
timeout 600 python3 bench.py --mode ba --steps 10 --warmup 3 --no-cpu-baseline > $O/r03_bench_final13682_ba6.json 2>/dev/null
timeout 600 python3 bench.py --workload synthetic-10k --steps 10 --warmup 3 --no-cpu-baseline > $O/r03_bench_synthetic10k.json 2>/dev/null
for f in ladybug1723 venice1778 sphere2500 final13682_hub final13682_iterative final13682_ba6 synthetic10k; do python3 - $f <<'PY'
import json, sys
try:
    b = json.load(open(f"gpurun_out/r03_bench_{sys.argv[1]}.json"))
    cb = b.get("cpu_baseline") or {}
    print(sys.argv[1], round(b["value"], 3), b["unit"], "| cpu", cb.get("value"), cb.get("cores"), (cb.get("sample") or "")[:60], "| pcg", b["config"].get("pcg_iterations_per_step"))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
