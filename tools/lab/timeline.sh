#!/bin/bash
# on the GPU box: timeline of one step -> gpurun_out/<tag>_timeline.csv
R=$GRAFT_REPO_ROOT
TAG=${1:-tl}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/prof -o k -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
python3 tools/kernel_timeline.py $O/prof/k_results.db -2 > $R/gpurun_out/${TAG}_timeline.csv
rm -rf $O/prof
tail -1 $R/gpurun_out/${TAG}_timeline.csv
