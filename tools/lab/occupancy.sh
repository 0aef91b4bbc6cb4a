#!/bin/bash
# per kernel of a .hip file: VGPRs, LDS bytes, and how many workgroups per CU each allows (MI355X: 512 VGPRs per SIMD lane, 160 KB of LDS, 32 wavefronts per CU):  tools/lab/occupancy.sh ccl.hip [name regex]
cd "$(dirname "$0")/../../malstroem_amd/csrc"
hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -S --cuda-device-only -o /tmp/chk.s $1 2>/dev/null
python3 - "$2" <<'PY'
import re,sys
t=open('/tmp/chk.s').read()
for m in re.finditer(r'\.group_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.max_flat_workgroup_size:\s+(\d+)\n\s+\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)', t):
    lds,wg,name,scr,vg=m.groups()
    if re.search(sys.argv[1], name):
        lds=int(lds); vg=int(vg); wg=int(wg)
        waves=wg//64
        by_v=min(8, 512//max(8,((vg+7)//8*8)))
        wg_v=by_v*4//waves
        wg_l=(160*1024)//lds if lds else 99
        print("%-52s lds %6d vgpr %3d wg %4d -> WGs/CU by vgpr %2d by lds %2d" % (name[-52:], lds, vg, wg, wg_v, wg_l))
PY
