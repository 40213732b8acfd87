#!/bin/bash
# tools/sweep_opt.sh <option> <v1> <v2> ...   -- bench.py stage times for each value of an implementation switch
opt=$1; shift
for v in "$@"; do
  python bench.py --steps 6 --warmup 2 --no-cpu-baseline --opt $opt=$v > /tmp/sweep_$v.json 2>/dev/null
  python3 - "$opt" "$v" <<'PY'
import json, sys
b = json.load(open(f"/tmp/sweep_{sys.argv[2]}.json")); st = b["stages_ms_per_step"]
print(sys.argv[1], sys.argv[2], "total", round(b["value"], 2), {k: round(x, 2) for k, x in st.items() if x > 0.3})
PY
done
