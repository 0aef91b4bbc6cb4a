"""One-off check (development helper, needs an MI355X): the overlapped band chain with 3 bands on a 6000 x 5000 fBm DEM
against the oracle run on the undivided raster -- rasters and merged per-label records.  python tools/bigband_check.py"""
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, threading, time
import oracle
from bench import fbm
from malstroem_amd.distributed import BandPipeline, ThreadComm
n=6144
dem = fbm(n, beta=2.0, seed=5)[:6000, :5000].copy()
nb=3
out=[None]*nb
def work(comm):
    p = BandPipeline(comm, dem.shape, device=0)
    p.upload_dem(dem[p.row0:p.row0+p.nrows])
    rec = p.run_chain()
    out[comm.rank] = {k: p.download(k) for k in ("filled","noflat","flowdir","accum","labels","watersheds")}
    out[comm.rank]["n"] = p.nlabels; out[comm.rank].update(rec)
    p.close()
ts=[threading.Thread(target=work,args=(c,)) for c in ThreadComm.world(nb)]
[t.start() for t in ts]; [t.join() for t in ts]
t0=time.time()
filled = oracle.fill_terrain(dem); sh,dg = oracle.minimum_safe_short_and_diag(dem); nf = oracle.fill_terrain_no_flats(dem,sh,dg)
fd = oracle.terrain_flowdirection(nf); acc = oracle.accumulated_flow(fd); lab,nl = oracle.connected_components(oracle.depths(filled,dem))
ws = lab.copy(); oracle.watersheds_from_labels(fd, ws, 0)
print("oracle s", round(time.time()-t0,1))
cat = lambda k: np.concatenate([o[k] for o in out])
for k,w in (("filled",filled),("noflat",nf),("flowdir",fd),("accum",acc),("labels",lab),("watersheds",ws)):
    print(k, np.array_equal(cat(k), w))
print("nlabels", out[0]["n"], nl)
cnt = np.concatenate([o["counts"]["records"] for o in out]); print("counts", np.array_equal(cnt, np.bincount(ws.ravel(), minlength=nl+1)[1:]))
pp = np.concatenate([o["pour"]["records"] for o in out]); op = oracle.label_max_index(acc, lab, nl)[1:]
print("pour", all(np.array_equal(pp[f], op[f]) for f in ("value","row","col")))
st = np.concatenate([o["stats"]["records"] for o in out]); os_ = oracle.label_stats(oracle.depths(filled,dem), lab, nl)[1:]
print("stats", all(np.array_equal(st[f], os_[f]) for f in ("min","max","count")), np.allclose(st["sum"], os_["sum"], rtol=1e-12, atol=0))
