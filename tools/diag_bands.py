"""Diagnostic: plain fill of the top DIAG_H rows of the 65536-wide bench DEM as 4 bands vs undivided; where do they differ?"""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import DemSource
from malstroem_amd.distributed import BandPipeline, ThreadComm
from malstroem_amd.pipeline import HydroPipeline
H, W, NB = int(os.environ.get("DIAG_H", 16384)), int(os.environ.get("DIAG_W", 65536)), 4
src = DemSource(65536, 2.0)
rows = lambda r0, n: np.ascontiguousarray(src.rows(r0, n)[:, :W])
pipes = [None] * NB
def body(comm):
    p = pipes[comm.rank] = BandPipeline(comm, (H, W), device=0)
    p.upload_dem(rows(p.row0, p.nrows))
    p.fill()
    print("band", comm.rank, "exchanges", p.exchanges, flush=True)
ts = [threading.Thread(target=body, args=(c,)) for c in ThreadComm.world(NB)]
[t.start() for t in ts]; [t.join() for t in ts]
with HydroPipeline((H, W)) as pipe:
    for r0 in range(0, H, 2048):
        pipe.upload_rows("dem", r0, rows(r0, min(2048, H - r0)))
    pipe.run("fill"); pipe.sync()
    for p in pipes:
        got = p.download("filled"); want = pipe.download_rows("filled", p.row0, p.nrows)
        d = np.argwhere(got != want)
        print("band row0", p.row0, "differing cells", len(d))
        if len(d):
            print("  local rows", d[:, 0].min(), d[:, 0].max(), "cols", d[:, 1].min(), d[:, 1].max(), " band>undivided:", int((got[d[:, 0], d[:, 1]] > want[d[:, 0], d[:, 1]]).sum()),
                  " band<undivided:", int((got[d[:, 0], d[:, 1]] < want[d[:, 0], d[:, 1]]).sum()))
            print("  first:", d[:5].tolist(), got[d[0, 0], d[0, 1]], want[d[0, 0], d[0, 1]])
            rr = np.bincount(d[:, 0] // 62, minlength=p.nrows // 62 + 1); cc = np.bincount(d[:, 1] // 62)
            print("  tile rows with diffs:", np.flatnonzero(rr).tolist()[:40]); print("  tile cols with diffs:", np.flatnonzero(cc).tolist()[:40])
for p in pipes:
    p.close()
