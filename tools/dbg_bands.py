import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, oracle, threading
from _cases import fixtures, fbm
from malstroem_amd.distributed import BandPipeline, ThreadComm
fx = fixtures(); dem = fx['dtm']
n = 2
out = [None]*n
def work(comm):
    p = BandPipeline(comm, dem.shape, device=0)
    p.upload_dem(dem[p.row0:p.row0+p.nrows]); p.fill()
    out[comm.rank] = (p.row0, p.download('filled'), dict(p.exchanges), p.band.get_int('fill_rounds'))
ts = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(n)]
[t.start() for t in ts]; [t.join() for t in ts]
got = np.concatenate([o[1] for o in out]); want = fx['filled']
bad = np.argwhere(got != want)
print('mismatch', len(bad), [o[2] for o in out], [o[3] for o in out], 'row0s', [o[0] for o in out])
print('rows', np.unique(bad[:,0])[:40])
for r,c in bad[:10]: print(r,c,got[r,c],want[r,c], dem[r,c])
