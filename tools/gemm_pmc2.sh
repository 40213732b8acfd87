#!/bin/bash
# Counters of the tile GEMM forms (tools/gemm_bench <tasks> <third> <rounds> <mode>): clock under load (GRBM_GUI_ACTIVE / duration / 8 XCDs),
# matrix-pipe busy cycles, where the waves wait.  One --pmc pass per counter group (kernel trace only).
#   tools/gemm_pmc2.sh <tag> [gemm_bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for ctrs in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1)); out=gpurun_out/pmc_${tag}_$i; rm -rf $out
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --kernel-include-regex "k_tile_gemm_nt" -d $out -o p --output-format csv -- ./tools/gemm_bench "$@" > gpurun_out/pmc_${tag}_$i.log 2>&1
  ls $out >/dev/null 2>&1 || tail -5 gpurun_out/pmc_${tag}_$i.log
  python3 - $out <<'PY'
import csv,glob,collections,sys
d=sys.argv[1]
cc=glob.glob(d+"/**/*counter_collection.csv",recursive=True)[0]
kt=glob.glob(d+"/**/*kernel_trace.csv",recursive=True)[0]
dur={}
for r in csv.DictReader(open(kt)): dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]), r["Kernel_Name"])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    ns,name=dur.get(r["Dispatch_Id"],(0,""))
    key=name.split("(")[0][-40:]
    agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[key]["_ns"].append(ns)
for k,v in agg.items():
    n=len(v["_ns"]); ns=sum(v["_ns"])/n
    print(k, f"dispatch-counter rows {n}, mean duration {ns/1e3:.1f} us")
    for c,vals in sorted(v.items()):
        if c=="_ns": continue
        m=sum(vals)/len(vals)
        extra=f"  -> {m/ns/8:.3f} GHz" if c=="GRBM_GUI_ACTIVE" else ""
        print(f"   {c:32s} mean={m:.5g}{extra}")
PY
done
