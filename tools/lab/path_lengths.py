"""On the GPU box after tools/lab/dump_rasters.py: distribution of the in-tile flow path lengths (64 x 64 tiles of the accumulation),
i.e. how many cells are still walking in doubling round k of accum_tile_kernel (a cell pushes in round k iff its path inside the
tile is at least 2**k steps long)."""
import numpy as np

n = int(open("/tmp/mlab/meta.txt").read().split()[0])
fd = np.fromfile("/tmp/mlab/flowdir.bin", dtype=np.uint8).reshape(n, n)
DR = np.array([-1, -1, 0, 1, 1, 1, 0, -1, 0])
DC = np.array([0, 1, 1, 1, 0, -1, -1, -1, 0])
rng = np.random.default_rng(0)
T = 64
hist = np.zeros(14, np.int64)
rows_active = np.zeros(14, np.int64)
ntile = 0
maxlen = []
for _ in range(300):
    ti, tj = rng.integers(0, n // T, 2)
    w = fd[ti * T:(ti + 1) * T, tj * T:(tj + 1) * T].astype(np.int64)
    rr, cc = np.mgrid[0:T, 0:T]
    code = np.minimum(w, 8)
    nr, nc = rr + DR[code], cc + DC[code]
    inside = (code < 8) & (nr >= 0) & (nr < T) & (nc >= 0) & (nc < T)
    nxt = np.where(inside, nr * T + nc, -1).ravel()
    # path length by pointer doubling
    L = np.zeros(T * T, np.int64)
    A = nxt.copy()
    step = 1
    Lk = (A >= 0).astype(np.int64)      # length so far (counts steps of size `step` taken)
    L = Lk.copy()
    for k in range(13):
        act = A >= 0
        hist[k] += act.sum()
        rows_active[k] += act.reshape(T, T).any(axis=1).sum()
        if not act.any():
            break
        A2 = np.where(act, np.where(A >= 0, A[np.maximum(A, 0)], -1), -1)
        A = A2
    ntile += 1
print("tiles sampled", ntile)
for k in range(13):
    print("round %2d: active cells %5.1f %%   wave-rows with an active cell %5.1f %%" % (k, 100.0 * hist[k] / (ntile * T * T), 100.0 * rows_active[k] / (ntile * T)))
