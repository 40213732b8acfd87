#!/bin/bash
# Counter evidence for the dominant kernel (k_schur_pairs) on the headline workload, three separate --pmc passes
# (kernel-trace only) over tools/schur_bench.py:   tools/profile_pairs.sh r02  ->  gpurun_out/<tag>_k_schur_pairs_counters.txt
TAG=${1:-r03}
XA=${PAIRS_ARGS:-}   # extra schur_bench arguments, e.g. PAIRS_ARGS="--variants 2"
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
O=gpurun_out/${TAG}_k_schur_pairs_counters.txt
{
echo "# rocprofv3 --kernel-trace --pmc <set> --kernel-include-regex k_schur_pairs -- python3 tools/schur_bench.py --forms ${PAIRS_FORM:-4} --iters 3"
echo "# per-dispatch means, final-13682 SelfCalibration, $(date -u +%F)"
echo "## SQ issue / wait"
tools/pmc_kernel.sh ${TAG}_sq "k_schur_pairs" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" --forms ${PAIRS_FORM:-4} --iters 3 $XA | grep -v "^final\|^form"
echo "## SQ instruction mix / LDS"
tools/pmc_kernel.sh ${TAG}_sq2 "k_schur_pairs" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" --forms ${PAIRS_FORM:-4} --iters 3 $XA | grep -v "^final\|^form"
echo "## L1 / L2"
tools/pmc_kernel.sh ${TAG}_tcp "k_schur_pairs" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" --forms ${PAIRS_FORM:-4} --iters 3 $XA | grep -v "^final\|^form"
echo "## kernel time without counters"
python3 tools/schur_bench.py --forms ${PAIRS_FORM:-4} --iters 5 $XA 2>&1 | grep "^form"
} > $O 2>&1
cat $O
