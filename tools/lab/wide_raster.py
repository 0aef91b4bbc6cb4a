"""lab: one context of H x W cells (same cell count, different row stride) through the whole chain, N steps -- for rocprofv3
--kernel-trace --stats: does a kernel's time depend on the raster's WIDTH (row stride a larger power of two)?
usage: python tools/lab/wide_raster.py H W [steps]"""
import sys
import numpy as np
sys.path.insert(0, ".")
from bench import fbm
from malstroem_amd.pipeline import HydroPipeline

H, W = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
n = max(H, W)
dem = fbm(n, beta=2.0, seed=42)[:H, :W]
dem = np.ascontiguousarray(dem)
p = HydroPipeline((H, W), device=0)
p.upload("dem", dem)
names = ["fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints"]
for _ in range(steps):
    p.run(*names)
    p.sync()
print(H, W, {s: round(p.stage_ms(s), 3) for s in names})
p.close()
