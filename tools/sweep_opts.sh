#!/bin/bash
# tools/sweep_opts.sh "<a=1 b=2>" "<a=3>" ...   -- bench.py stage times for each SET of implementation switches
for set in "$@"; do
  args=""; for kv in $set; do args="$args --opt $kv"; done
  timeout 180 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $args > /tmp/sweep_one.json 2>/dev/null || { echo "$set: failed/timeout"; continue; }
  python3 - "$set" <<'PY'
import json, sys
b = json.load(open("/tmp/sweep_one.json")); st = b["stages_ms_per_step"]
print(sys.argv[1], "| total", round(b["value"], 2), "factor", round(st["factor"], 3))
PY
done
