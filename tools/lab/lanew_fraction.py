"""On the GPU box: how many 62 x 62 tiles of the benchmark DEM hold FLAT cells of more than one binade class (the no-flats rounds relax
those with the full adjacency words and per-cell weights -- relax<LaneW> -- instead of the packed blocks)."""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R)
import bench
from malstroem_amd.pipeline import HydroPipeline
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
src = bench.DemSource(n, 2.0)
dem = src.rows(0, n)
with HydroPipeline((n, n)) as p:
    p.upload("dem", dem); p.run("fill"); p.sync()
    F = p.download("filled")
wet = F > dem                                   # (lake cells: the bulk of the flat cells)
b = F.view(np.uint32)
cls = ((b >> 23) & 0xff).astype(np.uint8)
T = 62
m = (n - 2) // T
Fi, Wi = cls[1:1 + m * T, 1:1 + m * T].reshape(m, T, m, T), wet[1:1 + m * T, 1:1 + m * T].reshape(m, T, m, T)
lo = np.where(Wi, Fi, 255).min(axis=(1, 3)); hi = np.where(Wi, Fi, 0).max(axis=(1, 3))
has = Wi.any(axis=(1, 3))
multi = has & (lo != hi)
print("tiles", m * m, "with lake cells", int(has.sum()), "of more than one class", int(multi.sum()), "= %.2f %%" % (100.0 * multi.sum() / max(has.sum(), 1)))
