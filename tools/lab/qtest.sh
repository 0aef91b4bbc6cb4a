#!/bin/bash
# on the GPU box: the flood's queue solve with different grid sizes (MHIP_PF_QGRID: workgroups launched; the first field of a
# configuration selected a 64-register build in round 4 and is ignored now), 30 steps each; prints ms/step, fill ms, the fill
# engines seen in the timed steps, visits
cd $GRAFT_REPO_ROOT
export MHIP_DEVELOPER=1 MHIP_POOL_POISON=0 MALSTROEM_BENCH_ALLOW_FALLBACK=1
for cfg in ${QCFGS:-3:768 4:768 4:1024 3:512 3:256}; do
  a=${cfg%%:*}; b=${cfg##*:}
  MHIP_PF_QWG=$a MHIP_PF_QGRID=$b timeout -k 10 200 python bench.py --steps ${QSTEPS:-30} --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('QWG=$a QGRID=$b', d['ms_per_step'], d['stages']['fill']['ms'], d['config'].get('engines_seen_in_timed_steps',{}).get('fill_algorithm'), d['config'].get('fill_visits'))"
done
