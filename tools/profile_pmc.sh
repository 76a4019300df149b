# HBM traffic of the headline step: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), summarised into profiles/ by tools/pmc_summary.py
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_fetch.log 2>&1
echo fetch done
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_write.log 2>&1
echo write done
cd $R
python3 tools/pmc_summary.py $(find gpurun_out/pmc_fetch -name '*_results.db' | head -1) $(find gpurun_out/pmc_write -name '*_results.db' | head -1) gpurun_out/pmc_hbm_traffic.json
