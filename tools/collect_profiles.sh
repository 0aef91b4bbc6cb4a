#!/bin/bash
# on the GPU box (gpurun -- bash tools/collect_profiles.sh): the profile set of a round under gpurun_out/<tag>/ (tag = first argument) -- PMC traffic table,
# rocprofv3 kernel statistics and the bench lines DESIGN.md quotes; copy the results into profiles/ afterwards
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r03a}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_f.log 2>&1
echo "pmc fetch done"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_w.log 2>&1
echo "pmc write done"
python3 tools/pmc_traffic.py $O/pmc_f/f_results.db $O/pmc_w/w_results.db 16384 > $O/${TAG}_pmc_hbm_traffic.json
cp $O/${TAG}_pmc_hbm_traffic.json profiles/${TAG}_pmc_hbm_traffic.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof -o k -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1
python3 tools/kernel_stats.py $O/prof/k_results.db > $O/${TAG}_kernel_stats.csv
echo "kernel stats done"
timeout -k 10 400 python3 bench.py 2>$O/bench.err | tail -1 > $O/${TAG}_bench16384.json
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 2>/dev/null | tail -1 > $O/${TAG}_bench16384_20steps.json
echo "bench done"
# one step as a timeline (every dispatch with its start, duration and the idle gap before it)
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/tl -o k -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline > $O/tl.log 2>&1
python3 tools/kernel_timeline.py $O/tl/k_results.db -2 > $O/${TAG}_step_timeline.csv
rm -rf $O/tl
timeout -k 10 300 python3 bench.py --beta 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench16384_beta3.json
timeout -k 10 300 python3 bench.py --config 2 2>/dev/null | tail -1 > $O/${TAG}_bench4096_config2.json
if [ -z "$QUICK" ]; then timeout -k 10 600 python3 bench.py --size 32768 --bands 4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench32768_4bands_1gpu.json; fi
# the launch shapes of the scaling run rehearsed on this ONE device (rows through the host transport, control plane in shared memory): 2 processes x 2 bands, 4 x 1
if [ -z "$QUICK" ]; then
  MALSTROEM_BAND_TRANSPORT=host timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29791 bench.py --gpus 2 --size 32768 --bands 4 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench32768_2procs_x_2bands_1gpu.json
  MALSTROEM_BAND_TRANSPORT=host timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29792 bench.py --gpus 4 --size 32768 --bands 4 --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/${TAG}_bench32768_4procs_x_1band_1gpu.json
  # where the wall clock of a band step goes (kernels of four bands side by side)
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/bt -o k -- python3 bench.py --size 32768 --bands 4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bt.log 2>&1
  python3 tools/band_attribution.py $O/bt/k_results.db 4 > $O/${TAG}_band_attribution.csv
  rm -rf $O/bt
fi
# the N = 1 point of the strong-scaling curve (BASELINE configs[3]: ONE 65536^2 DEM, 4 bands on this GPU): BIG=1 only -- DEM synthesis + 6 steps (the pool of recycled device blocks reaches its steady state in the second step)
if [ -n "$BIG" ]; then timeout -k 10 1000 python3 bench.py --gpus 1 --size 65536 --steps 4 --warmup 2 --no-cpu-baseline 2>$O/bench65536.err | tail -1 > $O/${TAG}_bench65536_4bands_1gpu.json; fi
rm -rf $O/pmc_f $O/pmc_w $O/prof
ls -la $O
