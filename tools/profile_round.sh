set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b2 -o b2 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_b2.log 2>&1
echo b2 done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dnet -o dnet -- python3 $R/bench.py --workload dnet --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_dnet.log 2>&1
echo dnet done
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_top -o top -- python3 $R/bench.py --workload topformer --head refiner --batch 4 --size 2048 --steps 8 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_top.log 2>&1
echo top done
cd $R
for n in b2 dnet top; do f=$(find gpurun_out/prof_$n -name '*_results.db' | head -1); python3 tools/rocpd_stats.py $f gpurun_out/${n}_kernel_stats.csv; tail -1 gpurun_out/prof_$n.log; done
