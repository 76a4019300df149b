# Everything profiles/r03_* is made from, in one gpurun call:  bash tools/profile_r03.sh
#   1. --kernel-trace --stats of the three round-1/2 workloads (tools/profile_round.sh) and of the TopFormer training step
#      -> gpurun_out/{b2,dnet,top,toptrain}_kernel_stats.csv
#   2. HBM bytes: --pmc FETCH_SIZE and --pmc WRITE_SIZE, each a run of its own (tools/profile_pmc.sh) -> gpurun_out/pmc_hbm_traffic.json
#   3. matrix-core utilisation: --pmc SQ set and --pmc GRBM_GUI_ACTIVE, each its own                   -> gpurun_out/mfma_util.json
#   4. the plain bench line of the same tree and box                                                   -> gpurun_out/bench_headline.json
# (no --pmc pass is combined with any trace other than --kernel-trace; the program follows `--` directly)
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
bash $R/tools/profile_round.sh
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_toptrain -o tt -- python3 $R/bench.py --workload topformer --train --head refiner --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_toptrain.log 2>&1
echo toptrain done
cd $R
python3 tools/rocpd_stats.py $(find gpurun_out/prof_toptrain -name '*_results.db' | head -1) gpurun_out/toptrain_kernel_stats.csv
tail -1 gpurun_out/prof_toptrain.log
bash $R/tools/profile_pmc.sh
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_mfma -o m -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_mfma.log 2>&1
echo mfma pass done
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_gui -o g -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_gui.log 2>&1
echo gui pass done
cd $R
python3 tools/mfma_util_summary.py $(find gpurun_out/pmc_mfma -name '*_results.db' | head -1) $(find gpurun_out/pmc_gui -name '*_results.db' | head -1) gpurun_out/b2_kernel_stats.csv gpurun_out/mfma_util.json
# the rocpd databases are tens of MiB each: only the summaries travel back
rm -rf gpurun_out/prof_b2 gpurun_out/prof_dnet gpurun_out/prof_top gpurun_out/prof_toptrain gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_mfma gpurun_out/pmc_gui
timeout -k 10 400 python3 bench.py > gpurun_out/bench_headline.json 2> gpurun_out/bench_headline.err
tail -c 600 gpurun_out/bench_headline.json
