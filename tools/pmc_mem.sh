# Memory-path counters of one python script, each --pmc set in a run of its own:  bash tools/pmc_mem.sh <tag> <script.py> [args] -> gpurun_out/pmcmem_<tag>.json
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
SCRIPT=$R/$1; shift
cd /tmp
i=0
for CS in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
          "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
          "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum" \
          "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
          "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $CS -d $R/gpurun_out/pmcmem_${TAG}_$i -o p -- python3 $SCRIPT "$@" > $R/gpurun_out/pmcmem_${TAG}_$i.log 2>&1
  echo "pass $i done"
done
cd $R
python3 tools/pmc_table.py gpurun_out/pmcmem_$TAG.json $(find gpurun_out/pmcmem_${TAG}_* -name '*_results.db') ${PMC_MATCH:+--match $PMC_MATCH}
rm -rf gpurun_out/pmcmem_${TAG}_[0-9]
