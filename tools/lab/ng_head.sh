#!/bin/bash
cd $GRAFT_REPO_ROOT
export MHIP_DEVELOPER=1
for v in "0,1" "1,2" "2,2" "4,2" "1,3" "2,3" "0,1"; do
  MHIP_NG_HEAD=$v python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python3 -c "
import json; d=json.loads(open('/tmp/b.json').read()); print('head $v:', d['ms_per_step'], 'noflat', d['stages']['noflat']['ms'])"
done
