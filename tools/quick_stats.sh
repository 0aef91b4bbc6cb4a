#!/bin/bash
# on the GPU box: kernel statistics of a short bench run -> gpurun_out/<tag>_kernel_stats.csv, and one bench line
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-quick}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
python3 tools/kernel_stats.py $O/prof/k_results.db > $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $O/prof
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $R/gpurun_out/${TAG}_bench.json
