#!/usr/bin/env python3
"""The structure sweep between "banded" and "photo collection": final-13682 with a fraction p of the landmarks drawn from a
global power-law camera popularity instead of the capture window (apex-solver_amd/synthetic.py, "-mix:<p>").  Per p (round 6):
what apexgpu_set_structure predicts for the two ways to the step and which one it builds (apexgpu_variant_costs), and three
bench.py lines through the plain Sparse surface -- the automatic choice, the direct factorisation forced ("auto_variant" 0; a
plan beyond the size limit is then refused) and the matrix-free PCG forced ("matrix_free_only" 1) -- so that the choice can be
held against min(direct, matrix-free).
  python tools/structure_sweep.py [--p 0.00001,0.0001,0.0003,0.001] [--scale 1.0] [--tag r06] [--steps 3]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("APEX_SYNTH_CACHE", "/tmp/apex_synth_cache")


def bench(workload, scale, extra, timeout=1500):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--scale", str(scale), "--no-cpu-baseline", "--no-other-variants",
           "--no-other-workloads", *extra]
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return None, f"no result within {timeout} s"
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if lines:
        return json.loads(lines[-1]), None
    err = [l for l in p.stderr.splitlines() if "Error" in l or "error" in l]
    return None, (err[-1] if err else p.stderr[-300:]).strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", default="0.00001,0.0001,0.0003,0.001")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--tag", default="r06")
    ap.add_argument("--steps", type=int, default=3)
    a = ap.parse_args()
    rows = []
    for p in ["0"] + a.p.split(","):
        wl = "final-13682" if p == "0" else f"final-13682-mix:{p}"
        row = dict(p=p, workload=wl)
        run = ["--steps", str(a.steps), "--warmup", "1"]
        for name, extra in (("auto", []), ("direct", ["--opt", "auto_variant=0"]), ("matrix_free", ["--opt", "matrix_free_only=1", "--variant", "implicit"])):
            b, err = bench(wl, a.scale, run + extra)
            if b is None:
                row[name + "_ms"] = None
                row[name + "_note"] = err
                continue
            c = b["config"]
            row[name + "_ms"] = round(b["value"], 2)
            if name == "auto":
                row.update(tile_rows=c.get("s_tile_rows"), tiles=c.get("s_tiles"), levels=c.get("etree_levels"), choice=c.get("variant_choice"),
                           predicted_direct_ms=round(c.get("predicted_direct_ms", float("nan")), 1),
                           predicted_matrix_free_ms=round(c.get("predicted_matrix_free_ms", float("nan")), 1), variant_used=c.get("variant_used"))
            if name == "matrix_free":
                row["pcg_iterations"] = b.get("pcg_iterations_per_step")
        ms = [row[k] for k in ("direct_ms", "matrix_free_ms") if row.get(k) is not None]
        if ms and row.get("auto_ms") is not None:
            row["auto_over_best"] = round(row["auto_ms"] / min(ms), 3)
        print(json.dumps(row), flush=True)
        rows.append(row)
    out = os.path.join(ROOT, "gpurun_out", f"{a.tag}_structure_sweep.json")
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
