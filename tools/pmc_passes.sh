# Counter passes over one python script (each --pmc set in a run of its own, as the pool requires):
#   bash tools/pmc_passes.sh <tag> <script.py> [script args]   -> gpurun_out/pmc_<tag>.json
# SQ set 1: where waves spend their time; SQ set 2: instruction mix / MFMA; TCC: HBM bytes (FETCH_SIZE x2 on gfx950).
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
SCRIPT=$R/$1; shift
cd /tmp
i=0
for CS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAVES" \
          "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
          "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $CS -d $R/gpurun_out/pmc_${TAG}_$i -o p -- python3 $SCRIPT "$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1
  echo "pass $i done"
done
cd $R
python3 tools/pmc_table.py gpurun_out/pmc_$TAG.json $(find gpurun_out/pmc_${TAG}_* -name '*_results.db') ${PMC_MATCH:+--match $PMC_MATCH}
