#!/bin/bash
# experiment: label branch start point x CU mask of the side streams
export MHIP_DEVELOPER=1
for start in 2 1 0; do
  for mask in none ffffffff 55555555 11111111 01010101; do
    if [ "$mask" = none ]; then unset MHIP_SIDE_CUMASK; else export MHIP_SIDE_CUMASK=$mask; fi
    MHIP_LABEL_START=$start python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('start $start mask $mask: %.2f ms/step ' % d['ms_per_step'], {k: v['ms'] for k, v in d['stages'].items()})"
  done
done
