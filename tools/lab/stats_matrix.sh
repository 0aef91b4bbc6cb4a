#!/bin/bash
# on the GPU box: stats_kernel alone (serial stages) over the development knobs MHIP_STATS_MODE x MHIP_STATS_GRID
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cd $R
export MHIP_DEVELOPER=1 MHIP_SERIAL=1
for mode in 0 1 2; do for grid in 2048 1536 1280 1024; do
  O=$R/gpurun_out/sm_${mode}_${grid}
  MHIP_STATS_MODE=$mode MHIP_STATS_GRID=$grid timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O -o k -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  echo "mode $mode grid $grid: $(python3 tools/kernel_stats.py $O/k_results.db | python3 -c "
import csv, sys
for r in csv.reader(sys.stdin):
    if r and 'stats_kernel<' in r[0]: print(r[0][:40], 'avg_us', r[3])
")"
  rm -rf $O
done; done
