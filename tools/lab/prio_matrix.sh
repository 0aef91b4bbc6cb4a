#!/bin/bash
# on the GPU box: step time over the stream priorities of the main stream (fill .. accumulation, pour points) and the side streams
cd $GRAFT_REPO_ROOT
export MHIP_DEVELOPER=1
for main in 1 0; do for side in 0 1; do
  MHIP_MAIN_PRIO=$main MHIP_SIDE_PRIO=$side python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python3 -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('main $main side $side:', d['ms_per_step'], {k:v['ms'] for k,v in d['stages'].items()})"
done; done
