"""CPU stand-in for ``malstroem_amd.distributed.HipBand`` (TEST INFRASTRUCTURE): same methods, NumPy/oracle
compute, so that the row-band PROTOCOL (halo swaps, activity all-reduce, global short/diag) can be exercised over
real ``torch.distributed``/gloo processes on a machine without a GPU.  Never imported by the product."""
import numpy as np

import oracle


class CpuBand(object):
    def __init__(self, H_global, W, row0, nrows, device=0, rank=0, size=1):
        self.Hg, self.W, self.row0, self.nrows = H_global, W, row0, nrows
        self.ht = 1 if row0 > 0 else 0
        self.hb = 1 if row0 + nrows < H_global else 0
        self.H = nrows + self.ht + self.hb
        self.r = {}
        self.kind_state = {}

    def close(self):
        pass

    def _dtype(self, name):
        return {"dem": np.float32, "filled": np.float32, "depths": np.float32, "noflat": np.float64, "flowdir": np.uint8}[name]

    def _raster(self, name):
        if name not in self.r:
            self.r[name] = np.zeros((self.H, self.W), dtype=self._dtype(name))
        return self.r[name]

    def upload(self, name, arr):
        self._raster(name)[self.ht:self.ht + self.nrows] = arr

    def download(self, name):
        return self.r[name][self.ht:self.ht + self.nrows].copy()

    def get_edge_row(self, name, side):
        return self._raster(name)[self.ht if side == 0 else self.ht + self.nrows - 1].copy()

    def set_halo_row(self, name, side, row):
        a = self._raster(name)
        i = 0 if side == 0 else self.H - 1
        changed = not np.array_equal(a[i], row, equal_nan=True)
        a[i] = row
        return changed

    def dem_minmax(self):
        d = self.r["dem"][self.ht:self.ht + self.nrows]
        return d.min(), d.max(), bool(np.isnan(d).any())

    # ---- fills: vectorised Jacobi to the local fixed point with frozen halo rows (schedule independent result)
    def _solve(self, kind):
        dem = self.r["dem"]
        w = self.r["filled" if kind == 0 else "noflat"]
        sh, dg = self.kind_state[kind]
        lo = 1
        hi = self.H - 1
        d = dem.astype(w.dtype)
        while True:
            p = np.pad(w, 1, constant_values=np.inf)
            n = lambda dr, dc: p[1 + dr:1 + dr + self.H, 1 + dc:1 + dc + self.W]
            if kind == 0:
                m = np.minimum.reduce([w, n(-1, -1), n(-1, 0), n(-1, 1), n(0, -1), n(0, 1), n(1, -1), n(1, 0), n(1, 1)])
            else:
                md = np.minimum.reduce([n(-1, -1), n(-1, 1), n(1, -1), n(1, 1)]) + dg
                me = np.minimum.reduce([n(-1, 0), n(1, 0), n(0, -1), n(0, 1)]) + sh
                m = np.minimum(np.minimum(md, me), w)
            new = np.maximum(m, d)
            new[:lo] = w[:lo]
            new[hi:] = w[hi:]
            new[:, 0] = w[:, 0]
            new[:, -1] = w[:, -1]
            if np.array_equal(new, w):
                return
            w[...] = new

    def fill_begin(self, kind, short=0.0, diag=0.0):
        self.kind_state[kind] = (short, diag)
        dem = self.r["dem"]
        w = self._raster("filled" if kind == 0 else "noflat")
        w[...] = np.inf
        w[:, 0] = dem[:, 0]
        w[:, -1] = dem[:, -1]
        if not self.ht:
            w[0] = dem[0]
        if not self.hb:
            w[-1] = dem[-1]
        self._solve(kind)
        return False

    def fill_batch(self, kind):
        self._solve(kind)
        return False

    def fill_halo_changed(self, kind, side):
        pass

    def fill_end(self, kind):
        if kind == 0:
            self._raster("depths")[...] = self.r["filled"] - self.r["dem"]

    def run_flowdir(self):
        z = self.r["noflat"]
        fd = oracle.terrain_flowdirection(z, edges_flow_outward=False)
        gr = np.arange(self.H) + (self.row0 - self.ht)
        maxr, maxc = self.Hg - 1, self.W - 1
        # flow.py:130-139 in global coordinates
        fd[gr == 0, :] = 0
        fd[gr == maxr, :] = 4
        fd[:, 0] = 6
        fd[:, maxc] = 2
        if (gr == 0).any():
            fd[gr == 0, 0] = 7
            fd[gr == 0, maxc] = 1
        if (gr == maxr).any():
            fd[gr == maxr, 0] = 5
            fd[gr == maxr, maxc] = 3
        self._raster("flowdir")[...] = fd

    def get_int(self, key):
        return 0
