import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from _cases import fbm
from malstroem_amd.pipeline import HydroPipeline
rng = np.random.default_rng(0)
def run(dem):
    with HydroPipeline(dem.shape) as pipe:
        pipe.upload("dem", dem)
        pipe.run("fill", "noflat")
        pipe.sync()
        return pipe.get_int("noflat_algorithm"), pipe.get_int("noflat_rounds")
dem = fbm(188, 250, beta=2.0, seed=5) + np.float32(3.0)
for it in range(12):
    # dirty the pool with other sizes in between
    if it % 3 == 1: run(fbm(700, 450, beta=3.0, seed=6) + np.float32(3.0))
    if it % 3 == 2: run((rng.random((188, 250)) * 1000).astype(np.float32))
    print(it, run(dem), flush=True)
