export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
timeout 900 python -m pytest tests/test_gpu_tri_dataflow.py tests/test_gpu_parity.py tests/test_gpu_pg_parity.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for o in 1 0; do
timeout 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --opt tri_dataflow=$o 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('flow=$o', d['value'], d['stages_ms_per_step']['tri_solve'], d['stages_ms_per_step']['factor'])"
done
timeout 300 python3 bench.py --workload sphere2500 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('sphere', d['value'], d['stages_ms_per_step'])"
timeout 300 python3 bench.py --workload final-13682-hub --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('hub', d['value'], d['stages_ms_per_step']['tri_solve'], d['stages_ms_per_step']['factor'])"
