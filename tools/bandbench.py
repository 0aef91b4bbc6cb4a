"""Time the band chain on ONE GPU (development helper): `nbands` bands of size x size each, stacked to a
(nbands*size) x size DEM, driven by threads over ThreadComm.  The bands share the GPU, so the wall time is roughly
nbands x (one band's time) + protocol overhead; compare with nbands x the single-raster chain.
    python tools/bandbench.py --size 8192 --bands 2"""
import argparse, sys, threading, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from _cases import fbm
from malstroem_amd.distributed import BandPipeline, ThreadComm

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--bands", type=int, default=2)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--chain", choices=["stages", "serial", "overlap"], default="stages", help="stage by stage, or BandPipeline.run_chain")
ap.add_argument("--profile", action="store_true", help="cProfile of rank 0's last repetition")
args = ap.parse_args()
tile = fbm(args.size, beta=2.0)
dem = np.concatenate([tile if k % 2 == 0 else tile[::-1] for k in range(args.bands)])
res = {}

def work(comm):
    p = BandPipeline(comm, dem.shape, device=0)
    p.upload_dem(dem[p.row0:p.row0 + p.nrows])
    for rep in range(args.reps):
        t = {}
        prof = None
        if args.profile and comm.rank == 0 and rep == args.reps - 1:
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        t_all = time.perf_counter()
        if args.chain == "stages":
            for name, fn in (("fill", p.fill), ("noflat", p.noflat), ("flowdir", p.flowdir), ("accum", p.accum), ("label", p.label), ("watershed", p.watershed)):
                t0 = time.perf_counter()
                fn()
                t[name] = round((time.perf_counter() - t0) * 1e3, 1)
        else:
            tm = {}
            p.run_chain(records=True, fetch_own=False, overlap=args.chain == "overlap", timings=tm)
            t.update({k: round(v, 1) for k, v in tm.items()})
        t["total"] = round((time.perf_counter() - t_all) * 1e3, 1)
        t["exchanges"] = dict(p.exchanges)
        res[(comm.rank, rep)] = t
        if prof is not None:
            import pstats
            prof.disable()
            pstats.Stats(prof).sort_stats("cumulative").print_stats(28)
    p.close()

ts = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(args.bands)]
[t.start() for t in ts]; [t.join() for t in ts]
for k in sorted(res):
    print(k, res[k])
