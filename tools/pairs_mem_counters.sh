#!/bin/bash
# Memory-path counters of the pair kernel (which unit the gathers of a chunk wait for): tools/pairs_mem_counters.sh <tag>
TAG=${1:-r06}
export APEX_SYNTH_CACHE=/tmp/apex_synth_cache
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG}_pairs_mem_counters.txt
{
echo "# available TA / TCP / TD / TCC counters (rocprofv3 -L)"
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC|SQ_INST_LEVEL|SQ_WAIT|SQ_INSTS_VMEM|SQ_ACTIVE_INST_VMEM|SQ_VMEM)[A-Za-z0-9_]*" | sort -u | tr '\n' ' '
echo
for set in "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN2_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_COALESCABLE_WAVEFRONT_sum" \
           "TCC_REQ_sum TCC_READ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  echo "## $set"
  tools/pmc_kernel.sh ${TAG}_mem "k_schur_pairs" "$set" --forms 4 --iters 2 2>&1 | grep -v "^final\|^form" | tail -8
done
} > $O 2>&1
cat $O
