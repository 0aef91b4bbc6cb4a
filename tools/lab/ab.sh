#!/bin/bash
# on the GPU box: A/B of a development knob, alternating runs:  bash tools/lab/ab.sh KNOB valueA valueB [pairs]
cd $GRAFT_REPO_ROOT
export MHIP_DEVELOPER=1
K=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do for v in $A $B; do
  env $K=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python3 -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('$K=$v:', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"
done; done
