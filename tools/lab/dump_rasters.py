"""On the GPU box: the benchmark DEM through the pipeline once, the rasters the tail kernels read dumped to /tmp/mlab/*.bin for
tools/lab/taillab.hip (development only)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bench import fbm                       # noqa: E402
from malstroem_amd.pipeline import HydroPipeline   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
out = "/tmp/mlab"
os.makedirs(out, exist_ok=True)
dem = fbm(n)
with HydroPipeline(dem.shape) as p:
    p.upload("dem", dem)
    p.run("fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints")
    p.sync()
    for name in ("depths", "labels", "watersheds", "accum", "flowdir"):
        a = p.download(name)
        a.tofile(os.path.join(out, name + ".bin"))
        print(name, a.dtype, a.shape, flush=True)
    nl = int(p.get_int("nlabels"))
with open(os.path.join(out, "meta.txt"), "w") as f:
    f.write("%d %d %d\n" % (n, n, nl))
print("nlabels", nl)
