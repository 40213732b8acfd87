#!/usr/bin/env python3
"""The structure sweep between "banded" and "hubs" (round 4): final-13682 with a fraction p of the landmarks drawn from a
global power-law camera popularity instead of the capture window (apex-solver_amd/synthetic.py, "-mix:<p>").  Per p:
the tile structure (host only), one bench.py line of the Cholesky variant -- with the Iterative and matrix-free runs that
follow its timed region -- or, where the fill makes S dense (the tile update list is refused above 80 M products = 4.8e14
flop), the reason and a bench.py line of the matrix-free variant at the reference's 500 / 1e-9.
  python tools/structure_sweep.py [--p 0.0001,0.001,0.01,0.05,0.2] [--scale 1.0] [--tag r04]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("APEX_SYNTH_CACHE", "/tmp/apex_synth_cache")


def bench(workload, scale, extra):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--scale", str(scale), "--no-cpu-baseline", *extra]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=1800)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if lines:
        return json.loads(lines[-1]), None
    err = [l for l in p.stderr.splitlines() if "Error" in l or "error" in l]
    return None, (err[-1] if err else p.stderr[-300:]).strip()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", default="0.0001,0.001,0.01,0.05,0.2")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--tag", default="r04")
    a = ap.parse_args()
    import apex_solver_amd as pkg

    rows = []
    for p in ["0"] + a.p.split(","):
        wl = "final-13682" if p == "0" else f"final-13682-mix:{p}"
        d = pkg.datasets.load_named(wl, a.scale)[0]
        hs = pkg.capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx)
        row = dict(p=p, workload=d.name, tile_rows=int(hs["tile_rows"]), tiles=int(hs["tiles"]), touched=int(hs["touched_tiles"]),
                   levels=int(hs["etree_levels"]), border_cameras=int(hs["hub_cameras"]), host_setup_s=round(hs["s_total"], 2))
        dense = row["tile_rows"] * (row["tile_rows"] + 1) // 2
        row["fill_of_dense"] = round(row["tiles"] / dense, 3)
        del d
        b, err = bench(wl, a.scale, ["--steps", "5", "--warmup", "2"])
        if b:
            st = b["stages_ms_per_step"]
            row.update(cholesky_ms=round(b["value"], 2), factor_ms=round(st["factor"], 2), pairs_ms=round(st["schur_scatter"], 2), tri_ms=round(st["tri_solve"], 2),
                       iterative_ms=round(b.get("iterative_ms", float("nan")), 1), fallback_ms_implicit=round(b.get("fallback_ms_implicit", float("nan")), 1),
                       implicit_pcg_iterations=b.get("other_variants", {}).get("fallback_ms_implicit", {}).get("pcg_iterations"))
        else:
            row["cholesky"] = "not run: " + err
            b2, err2 = bench(wl, a.scale, ["--variant", "implicit", "--steps", "3", "--warmup", "1"])
            if b2:
                row.update(fallback_ms_implicit=round(b2["value"], 1), implicit_pcg_iterations=b2.get("pcg_iterations_per_step"))
            else:
                row["implicit"] = "failed: " + err2
        print(json.dumps(row), flush=True)
        rows.append(row)
    out = os.path.join(ROOT, "gpurun_out", f"{a.tag}_structure_sweep.json")
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
