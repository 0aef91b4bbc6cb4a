#!/bin/bash
# on the GPU box: a development knob over several values, alternating runs:  bash tools/lab/abn.sh KNOB "v1 v2 v3" [rounds] [bench args]
cd $GRAFT_REPO_ROOT
export MHIP_DEVELOPER=1
K=$1; VALS=$2; N=${3:-2}; shift $(( $# < 3 ? $# : 3 ))
for i in $(seq $N); do for v in $VALS; do
  env $K=$v python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python3 -c "
import json; d=json.loads(open('/tmp/b.json').read()); i=d.get('config',{}); print('$K=$v:', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, {k:i.get(k) for k in ('noflat_rounds','noflat_visits','fill_visits')})"
done; done
