# The structure sweep (round 4): final-13682 with a fraction p of the landmarks drawn from a global power-law camera popularity.
# Per p: tiles / levels / factor ms of the Cholesky variant, and the matrix-free variant at the reference's 500 / 1e-9.
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
O=gpurun_out; T=${1:-r04}
for p in 0.01 0.05 0.2; do
  timeout 1500 python3 bench.py --workload final-13682-mix:$p --steps 5 --warmup 2 --no-cpu-baseline > $O/${T}_bench_final13682_mix_$p.json 2> $O/${T}_mix_$p.err
  python3 - $p $O/${T}_bench_final13682_mix_$p.json <<'PY'
import json, sys
try:
    b = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); c = b["config"]; st = b["stages_ms_per_step"]
    print("mix p=%s: %.1f ms/LM iter | tiles %d (touched %d) levels %d border cameras %d | factor %.1f pairs %.1f tri %.2f | iterative %.1f ms | matrix-free fallback %.1f ms (%s PCG iterations) | setup %.2f s" % (
        sys.argv[1], b["value"], c["s_tiles"], c["s_tiles_touched"], c["etree_levels"], c["border_cameras"], st["factor"], st["schur_scatter"], st["tri_solve"],
        b.get("iterative_ms", float("nan")), b.get("fallback_ms_implicit", float("nan")), b.get("other_variants", {}).get("fallback_ms_implicit", {}).get("pcg_iterations"), b["setup_s"]))
except Exception as e:
    print("mix p=%s FAILED" % sys.argv[1], e, open(sys.argv[2].replace("_bench_final13682_mix_", "_mix_").replace(".json", ".err")).read()[-600:])
PY
done 2>&1 | tee $O/${T}_structure_sweep.txt
