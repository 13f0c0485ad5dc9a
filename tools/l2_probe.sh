#!/bin/bash
# Dev tool (GPU): where the global -> CU path of the headline kernels stalls.  Separate rocprofv3 --pmc passes over a short bench run
# (one batch in flight; the TA / TCP blocks take two counters per pass: more abort the run with "exceeds the capabilities of the
# hardware"), condensed per kernel by tools/l2_probe_summary.py into gpurun_out/l2probe/summary.txt.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/l2probe; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
PASSES=${L2_PASSES:-all}
for c in "TCC_HIT_sum TCC_MISS_sum TCC_READ_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum" \
         "TA_TA_BUSY GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
         "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ" "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD" \
         "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_REQUEST" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1)); d=$O/pass$i
  if [ "$PASSES" != all ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  timeout 60 rocprofv3 --pmc $c --output-format csv -d $d -- python $R/bench.py --steps 6 --warmup 2 --slots 1 --preheat-ms 0 --no-cpu-baseline > $d.log 2>&1
  grep -i -E "error|invalid|not supported|unable" $d.log | head -3
done
python $R/tools/l2_probe_summary.py $O | tee $O/summary.txt
