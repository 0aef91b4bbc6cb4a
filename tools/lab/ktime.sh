#!/bin/bash
# on the GPU box: per-kernel statistics of a short bench run:  bash tools/lab/ktime.sh OUT.csv [bench args]   (development knobs from the environment)
R=$GRAFT_REPO_ROOT; OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf /tmp/kt && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > /tmp/kt.log 2>&1
python3 tools/kernel_stats.py /tmp/kt/k_results.db > $OUT
