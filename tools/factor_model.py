"""Offline model of the tile Cholesky of a workload (host only, no GPU): tile structure in the solver's camera order, symbolic
fill, elimination-tree levels, tile products per level, and a time estimate
    level time = max(latency of potrf + panel + update launches, products * 2 * 144^3 / GEMM rate).
Usage: python tools/factor_model.py [workload] [scale] [--nd LEAF]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import apex_solver_amd as pkg  # noqa: E402
from apex_solver_amd import capi  # noqa: E402


def tile_graph(d, cmap, dc=9):
    per = 144 // dc
    t = cmap[d.cam_idx] // per
    nt = (d.n_cam * dc + 143) // 144
    order = np.argsort(d.pt_idx, kind="stable")
    pt = d.pt_idx[order]; tt = t[order]
    ptr = np.searchsorted(pt, np.arange(d.n_pt + 1))
    present = np.zeros((nt, nt), dtype=bool)
    # unique tiles per landmark -> clique
    k = np.diff(ptr)
    for kk in np.unique(k):
        if kk < 1:
            continue
        idx = np.nonzero(k == kk)[0]
        rows = tt[ptr[idx][:, None] + np.arange(kk)[None, :]]
        for a in range(kk):
            for b in range(kk):
                present[rows[:, a], rows[:, b]] = True
    return present


def symbolic(present):
    nt = present.shape[0]
    cols = [set(np.nonzero(present[j + 1:, j])[0] + j + 1) for j in range(nt)]
    parent = [-1] * nt
    for j in range(nt):
        if cols[j]:
            p = min(cols[j]); parent[j] = p
            cols[p] |= (cols[j] - {p})
    level = [0] * nt
    for j in range(nt):
        if parent[j] >= 0:
            level[parent[j]] = max(level[parent[j]], level[j] + 1)
    return cols, parent, level


def report(cols, level, rate=47e12, lat=100e-6):
    nt = len(cols)
    nl = max(level) + 1
    prod = np.zeros(nl); ncol = np.zeros(nl, dtype=int)
    for j in range(nt):
        r = len(cols[j])
        prod[level[j]] += r + r * (r + 1) // 2
        ncol[level[j]] += 1
    t = np.maximum(lat, prod * 2 * 144 ** 3 / rate)
    print(f"tiles {sum(len(c) for c in cols) + nt}  levels {nl}  products {int(prod.sum())}  model {t.sum() * 1e3:.2f} ms "
          f"(work {prod.sum() * 2 * 144 ** 3 / rate * 1e3:.2f} ms, latency {nl * lat * 1e3:.2f} ms)")
    return prod, ncol


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="final-13682")
    ap.add_argument("scale", nargs="?", type=float, default=1.0)
    ap.add_argument("--nd", type=int, default=1)
    ap.add_argument("--levels", action="store_true")
    a = ap.parse_args()
    d = pkg.synthetic.make_named(a.workload, a.scale)
    hs = capi.host_structure(d.n_cam, d.n_pt, d.cam_idx, d.pt_idx, nested_dissection=a.nd)
    present = tile_graph(d, hs["cmap"])
    cols, parent, level = symbolic(present)
    prod, ncol = report(cols, level)
    if a.levels:
        for lv in range(len(prod)):
            print(lv, ncol[lv], int(prod[lv]))
