"""On the GPU box: in how many 64 x 64 windows (62 x 62 tiles + ring) of the benchmark DEM is every cell that is NOT adjacent to a
flat cell next to it a pure wall -- i.e. no source cell that feeds a flat of its own level AND touches a flat cell of a lower level
(a spill cell right next to the next lake).  Windows without such a cell could relax without per-direction adjacency bits."""
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
del dem
H = W = n
C = F[1:-1, 1:-1]
nb = [F[1 + dr:H - 1 + dr, 1 + dc:W - 1 + dc] for dr in (-1, 0, 1) for dc in (-1, 0, 1) if (dr, dc) != (0, 0)]
lower = np.zeros(C.shape, bool); equal = np.zeros(C.shape, bool)
for x in nb:
    lower |= x < C
    equal |= x == C
flat = np.zeros((H, W), bool)
flat[1:-1, 1:-1] = ~lower & equal                  # a flat cell: no lower neighbour, some neighbour on its level
srcc = np.zeros((H, W), bool)
srcc[1:-1, 1:-1] = lower
del lower, equal
has_eq = np.zeros(C.shape, bool); has_low = np.zeros(C.shape, bool)
for dr in (-1, 0, 1):
    for dc in (-1, 0, 1):
        if (dr, dc) == (0, 0):
            continue
        x = F[1 + dr:H - 1 + dr, 1 + dc:W - 1 + dc]; fx = flat[1 + dr:H - 1 + dr, 1 + dc:W - 1 + dc]
        has_eq |= fx & (x == C)
        has_low |= fx & (x < C)
mixed = np.zeros((H, W), bool)
mixed[1:-1, 1:-1] = srcc[1:-1, 1:-1] & has_eq & has_low
print("cells", H * W, "flat", int(flat.sum()), "sources next to a flat of their level", int((srcc[1:-1, 1:-1] & has_eq).sum()), "mixed", int(mixed.sum()))
T = 62
m = (n - 2) // T
cs = np.zeros((H + 1, W + 1), np.int64); cs[1:, 1:] = mixed.cumsum(0).cumsum(1)
fs = np.zeros((H + 1, W + 1), np.int64); fs[1:, 1:] = flat.cumsum(0).cumsum(1)
i = np.arange(m) * T
r0, r1 = i, np.minimum(i + 64, H)
def boxsum(S, a0, a1, b0, b1):
    return S[a1][:, b1] - S[a0][:, b1] - S[a1][:, b0] + S[a0][:, b0]
mx = boxsum(cs, r0, r1, r0, r1)
ft = boxsum(fs, r0 + 1, np.minimum(r0 + 63, H), r0 + 1, np.minimum(r0 + 63, W))
act = ft > 0
print("tiles", m * m, "with flat cells", int(act.sum()), "with a mixed source in the window", int((act & (mx > 0)).sum()),
      "= %.2f %%" % (100.0 * (act & (mx > 0)).sum() / max(act.sum(), 1)))
