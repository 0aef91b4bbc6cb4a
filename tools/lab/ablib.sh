#!/bin/bash
# on the GPU box: two builds of the library, alternating runs on one box:  bash tools/lab/ablib.sh path/to/old.so [rounds] [bench args]
cd $GRAFT_REPO_ROOT
OLD=$1; N=${2:-3}; shift $(( $# < 2 ? $# : 2 ))
for i in $(seq $N); do for v in old new; do
  if [ $v = old ]; then export MALSTROEM_HIP_LIB=$GRAFT_REPO_ROOT/$OLD; else unset MALSTROEM_HIP_LIB; fi
  python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | tail -1 > /tmp/b.json
  python3 -c "
import json; d=json.loads(open('/tmp/b.json').read()); i=d.get('config',{}); print('$v:', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()}, i.get('fill_visits'), i.get('engines_seen_in_timed_steps',{}).get('fill_algorithm'))"
done; done
