"""Diagnostic: fill of the top 16384 rows of the 65536-wide bench DEM by the priority-flood and by the iterative schedule
(two processes: MHIP_FILL is read once), rows [r0, r1) compared; for differing cells the fixed-point equation is evaluated."""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
H, W = int(os.environ.get("DIAG_H", 16384)), 65536
if len(sys.argv) > 1:
    from bench import DemSource
    from malstroem_amd.pipeline import HydroPipeline
    src = DemSource(W, 2.0)
    with HydroPipeline((H, W)) as pipe:
        for r0 in range(0, H, 2048):
            pipe.upload_rows("dem", r0, src.rows(r0, min(2048, H - r0)))
        pipe.run("fill"); pipe.sync()
        print(sys.argv[1], "algorithm", pipe.get_int("fill_algorithm"), "launches", pipe.get_int("fill_launches"), flush=True)
        np.save("/dev/shm/diag_%s.npy" % sys.argv[1], pipe.download("filled"))
        if sys.argv[1] == "pf":
            np.save("/dev/shm/diag_dem.npy", pipe.download("dem"))
    sys.exit(0)
for tag, env in (("pf", {}), ("it", {"MHIP_FILL": "iterative"})):
    subprocess.check_call([sys.executable, __file__, tag], env=dict(os.environ, **env))
a, b, dem = (np.load("/dev/shm/diag_%s.npy" % t, mmap_mode="r") for t in ("pf", "it", "dem"))
bad = []
for r0 in range(0, H, 1024):
    d = np.argwhere(a[r0:r0 + 1024] != b[r0:r0 + 1024])
    if len(d):
        d[:, 0] += r0
        bad.append(d)
bad = np.concatenate(bad) if bad else np.zeros((0, 2), int)
print("differing cells:", len(bad))
if len(bad):
    print("rows", bad[:, 0].min(), bad[:, 0].max(), "cols", bad[:, 1].min(), bad[:, 1].max())
    for r, c in bad[:10]:
        for name, f in (("pf", a), ("it", b)):
            nb = np.array(f[r - 1:r + 2, c - 1:c + 2]); nb[1, 1] = np.inf
            print(name, (r, c), "F", f[r, c], "dem", dem[r, c], "min nbr", nb.min(), "fixed point:", f[r, c] == max(dem[r, c], nb.min()))
    print("pf > it:", int((a[bad[:, 0], bad[:, 1]] > b[bad[:, 0], bad[:, 1]]).sum()), "pf < it:", int((a[bad[:, 0], bad[:, 1]] < b[bad[:, 0], bad[:, 1]]).sum()))
for t in ("pf", "it", "dem"):
    os.remove("/dev/shm/diag_%s.npy" % t)
