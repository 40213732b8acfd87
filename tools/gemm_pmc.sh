#!/bin/bash
# clock under load: GRBM_GUI_ACTIVE (GPU cycles while busy) against the kernel durations of the trace
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for prog in "./tools/gemm_bench 4096" "./tools/wave_placement_bench"; do
rm -rf gpurun_out/pmc_clk
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-include-regex "k_tile_gemm_nt|k_mfma" -d gpurun_out/pmc_clk -o p --output-format csv -- $prog > gpurun_out/pmc_clk.log 2>&1
python3 - <<'PY'
import csv,glob,collections
cc=glob.glob("gpurun_out/pmc_clk/**/*counter_collection.csv",recursive=True)[0]
kt=glob.glob("gpurun_out/pmc_clk/**/*kernel_trace.csv",recursive=True)[0]
dur={}
for r in csv.DictReader(open(kt)): dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["Kernel_Name"][:40])
agg=collections.defaultdict(dict)
for r in csv.DictReader(open(cc)): agg[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
for d,(ns,name) in list(dur.items())[:40]:
    if d in agg and "GRBM_GUI_ACTIVE" in agg[d]:
        g=agg[d]["GRBM_GUI_ACTIVE"]
        print(f"{name:40s} {ns/1e3:9.1f} us  GUI_ACTIVE {g:.4g}  -> {g/ns:.2f} GHz x(instances?)")
PY
done
