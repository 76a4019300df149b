# per-kernel durations of one python script:  bash tools/stats_run.sh <tag> <script.py> [args]  -> gpurun_out/<tag>_kernel_stats.csv
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
SCRIPT=$R/$1; shift
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats_$TAG -o s -- python3 $SCRIPT "$@" > $R/gpurun_out/stats_$TAG.log 2>&1
cd $R
python3 tools/rocpd_stats.py $(find gpurun_out/stats_$TAG -name '*_results.db' | head -1) gpurun_out/${TAG}_kernel_stats.csv
head -12 gpurun_out/${TAG}_kernel_stats.csv
