"""Timeline of ONE factorisation from a rocprofv3 kernel trace: per dependent group potrf / gemm start, end, gaps.
usage: python tools/factor_timeline.py <kernel_trace.csv> [which factorisation, default last]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    short = ("flow" if "factor_flow" in n else "potrf" if "potrf" in n else "gemm_small_strip" if "small_strip" in n else "gemm_small" if "gemm_nt_small" in n else
             "gemm" if "tile_gemm_nt" in n else "rows" if ("schur_rows" in n or "schur_pairs" in n) else "tri" if ("tri_step" in n or "tri_fwd_flow" in n) else None)
    if short:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, int(r["Queue_Id"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
ev.sort()
# factorisations = spans between a 'rows' kernel and the first 'tri' kernel after it
spans = []
i = 0
while i < len(ev):
    if ev[i][2] == "rows":
        j = i + 1
        while j < len(ev) and ev[j][2] != "tri" and ev[j][2] != "rows": j += 1
        if any(x[2] in ("potrf", "flow") for x in ev[i + 1:j]): spans.append((i + 1, j))   # (a cost-only pass has no factorisation)
        i = j
    else:
        i += 1
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(spans) - 1
a, b = spans[which]
f = ev[a:b]
t0 = f[0][0]
print("factorisation", which, "kernels", len(f), "span %.2f ms" % ((max(e[1] for e in f) - t0) / 1e6))
tot = {}
for s, e, k, q, g in f: tot[k] = tot.get(k, 0) + (e - s)
print("busy per kernel kind (ms):", {k: round(v / 1e6, 2) for k, v in tot.items()})
# critical path view: potrf launches and what happens between consecutive potrfs
for x in f:
    if x[2] == "flow": print("dataflow launch of the top groups: start %.1f us, duration %.1f us, %d workgroups" % ((x[0] - t0) / 1e3, (x[1] - x[0]) / 1e3, x[4]))
pot = [x for x in f if x[2] == "potrf"]
print("potrf launches", len(pot), "sum %.2f ms" % (sum(e - s for s, e, *_ in pot) / 1e6))
print("%4s %8s %8s %8s  | between this potrf's end and the next potrf's start: main-queue kernels" % ("lv", "start", "dur", "wgs"))
mainq = pot[0][3]
for i, (s, e, k, q, g) in enumerate(pot):
    nxt = pot[i + 1][0] if i + 1 < len(pot) else max(x[1] for x in f)
    between = [x for x in f if x[0] >= e - 1000 and x[0] < nxt and x[2] != "potrf"]
    desc = " ".join("%s[q%d,%dwg,%.0fus@+%.0f]" % (x[2].replace("gemm_", "g_"), x[3], x[4], (x[1] - x[0]) / 1e3, (x[0] - e) / 1e3) for x in between[:6])
    print("%4d %8.1f %8.1f %8d  | gap to next potrf %.0f us: %s" % (i, (s - t0) / 1e3, (e - s) / 1e3, g, (nxt - e) / 1e3, desc))
