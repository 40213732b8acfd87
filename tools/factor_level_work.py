#!/usr/bin/env python3
"""Tile products of the factorisation per elimination-tree level (host only) beside the measured period of each level launch
(potrf start to potrf start, profiles/rNN_factor_timeline.txt): where the factorisation is at the tile GEMM's rate and where not.
  python tools/factor_level_work.py [workload=final-13682] [timeline=profiles/r05_factor_timeline.txt] [rate TF/s=52]"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("APEX_SYNTH_CACHE", "/tmp/apex_synth_cache")
import numpy as np
import scipy.sparse as sp
import apex_solver_amd as pkg

name = sys.argv[1] if len(sys.argv) > 1 else "final-13682"
timeline = sys.argv[2] if len(sys.argv) > 2 else "profiles/r05_factor_timeline.txt"
rate = float(sys.argv[3]) * 1e12 if len(sys.argv) > 3 else 52e12
d = pkg.datasets.load_named(name, 1.0)[0]
hs = pkg.capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx)
cpt = 16
nt = int(hs["tile_rows"])
tile = (hs["cmap"][d.cam_idx] // cpt).astype(np.int64)          # tiles in the solver's final camera order
A = sp.csr_matrix((np.ones(len(tile), dtype=np.int32), (d.pt_idx.astype(np.int64), tile)), shape=(d.n_pt, nt))
G = (A.T @ A).tocoo()
cols = [set() for _ in range(nt)]
for i, j in zip(G.row, G.col):
    if i > j: cols[j].add(int(i))
parent = [-1] * nt
for k in range(nt):
    if cols[k]:
        p = min(cols[k]); parent[k] = p
        cols[p] |= cols[k] - {p}
lvl = [0] * nt
for k in range(nt):
    if parent[k] >= 0: lvl[parent[k]] = max(lvl[parent[k]], lvl[k] + 1)
nl = max(lvl) + 1
fl = np.zeros(nl); ncol = np.zeros(nl, dtype=int)
for k in range(nt):
    r = len(cols[k])
    fl[lvl[k]] += (45 / 81 * r + r * (r + 1) / 2) * 2 * 144 ** 3      # panel solves skip the zero blocks of L^-1; updates
    ncol[lvl[k]] += 1
starts, flow = [], None
for line in open(timeline):
    m = re.match(r"\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+\|", line)
    if m: starts.append(float(m.group(2)))
    m = re.search(r"dataflow launch of the top groups: start ([\d.]+) us, duration ([\d.]+) us", line)
    if m: flow = (float(m.group(1)), float(m.group(2)))
print(f"# {name}: {nt} tile columns, {sum(len(c) for c in cols) + nt} tiles of L, {nl} levels, {fl.sum() / 1e9:.1f} GFLOP of tile products; rate {rate / 1e12:.0f} TF/s; periods from {timeline}")
print(f"# {'level':>5} {'cols':>5} {'GFLOP':>8} {'cum %':>6} {'at rate us':>10} {'measured us':>11}")
cum = 0.0
regions = {}
for l in range(nl):
    cum += fl[l]
    per = starts[l + 1] - starts[l] if l + 1 < len(starts) else (flow[0] - starts[l] if flow and l + 1 == len(starts) else None)
    reg = "bulk (levels 0-8)" if l <= 8 else ("middle" if per is not None else "top (one dataflow launch)")
    a = regions.setdefault(reg, [0.0, 0.0]); a[0] += fl[l] / rate * 1e6; a[1] += per or 0.0
    print(f"  {l:5d} {ncol[l]:5d} {fl[l] / 1e9:8.2f} {100 * cum / fl.sum():6.1f} {fl[l] / rate * 1e6:10.0f} {per if per is not None else float('nan'):11.0f}")
if flow: regions["top (one dataflow launch)"][1] = flow[1]
print("# region: tile products at the GEMM's rate / measured (the work of a level also runs beside later levels: compare the sums)")
for k, (w, t) in regions.items():
    print(f"#   {k:28s} {w / 1e3:6.2f} ms / {t / 1e3:6.2f} ms")
