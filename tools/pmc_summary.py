#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), corrected as
MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE tallies 128-byte requests at 64 B, so
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (the counters are in KB)."""
import collections
import csv
import json
import re
import sys


def load(path, name):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    return agg


def main():
    f = load(sys.argv[1], "FETCH_SIZE"); w = load(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "workload": "final-13682 synthetic, selfcal (d_c=9), sparse variant, 1 GPU",
           "unit_note": "counter values are KB; corrected = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)",
           "kernels": {}}
    for k in sorted(f, key=lambda k: -f[k][0]):
        fk = f[k][0] / max(f[k][1], 1); wk = w[k][0] / max(w[k][1], 1) if k in w else 0.0
        out["kernels"][k] = {"FETCH_SIZE_KB_per_launch": fk, "launches_fetch": f[k][1], "WRITE_SIZE_KB_per_launch": wk,
                             "launches_write": w[k][1] if k in w else 0,
                             "hbm_bytes_per_launch_corrected": (2 * fk + wk) * 1024, "hbm_bytes_per_launch_raw": (fk + wk) * 1024}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
