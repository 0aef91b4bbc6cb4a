"""CPU stand-in for ``malstroem_amd.distributed.HipBand`` (TEST INFRASTRUCTURE): same methods, NumPy/oracle
compute, so that the row-band PROTOCOL (halo swaps, activity all-reduce, global short/diag) can be exercised over
real ``torch.distributed``/gloo processes on a machine without a GPU.  Never imported by the product."""
import numpy as np

import oracle


class CpuBand(object):
    def __init__(self, H_global, W, row0, nrows, device=0, rank=0, size=1, unique_id=None):
        self.Hg, self.W, self.row0, self.nrows = H_global, W, row0, nrows
        self.rank, self.size = rank, size
        # stand-in for the library's RCCL communicator: the "unique id" carries a TCP port and every band joins a
        # SocketComm on it (a collective call, like ncclCommInitRank), so BandPipeline's in-library transport path
        # (band.exchange_halo / band.allreduce_max) runs on CPU boxes too
        self.has_comm = False
        self._comm = None
        if unique_id is not None and size > 1:
            import struct
            from malstroem_amd.distributed import SocketComm
            self._comm = SocketComm(rank, size, "127.0.0.1", struct.unpack("<i", unique_id[:4])[0], timeout_s=120)
            self.has_comm = True
        self.ht = 1 if row0 > 0 else 0
        self.hb = 1 if row0 + nrows < H_global else 0
        self.H = nrows + self.ht + self.hb
        self.r = {}
        self.kind_state = {}

    def close(self):
        for c in (self._comm, getattr(self, "_comm_side", None)):
            if c is not None:
                c.close()

    # stand-ins for the side communicator / side stream bracket of the library (mhip_ctx_comm_add_side, mhip_ctx_side_begin)
    has_side_comm = False

    def add_side_comm(self, unique_id):
        import struct
        from malstroem_amd.distributed import SocketComm
        self._comm_side = SocketComm(self.rank, self.size, "127.0.0.1", struct.unpack("<i", unique_id[:4])[0], timeout_s=120)
        self.has_side_comm = True

    def side_begin(self):
        import threading
        self._side_thread = threading.get_ident()

    def side_end(self):
        self._side_thread = None

    def exchange_edge_rows(self, name):
        import threading
        comm = self._comm_side if getattr(self, "_side_thread", None) == threading.get_ident() else self._comm
        has_up, has_down = self.rank > 0, self.rank < self.size - 1
        return comm.exchange_rows(self.get_edge_row(name, 0) if has_up else None, self.get_edge_row(name, 1) if has_down else None)

    def fill_certify(self, kind):
        if getattr(self, "_attached", False):      # attached to an upper bound: the first "certification" is the relaxation itself
            self._attached = False
            before = self.r["noflat" if kind else "filled"].copy()
            self._solve(kind)
            return not np.array_equal(before, self.r["noflat" if kind else "filled"])
        return False      # the stand-in iterates whole-raster sweeps to a fixed point: nothing to certify

    @staticmethod
    def new_unique_id():
        import os
        import struct
        port = int(os.environ.get("CPUBAND_COMM_PORT", "0")) or (31000 + os.getpid() % 2000)
        CpuBand._ids = getattr(CpuBand, "_ids", 0) + 1          # (every id = another port: the side communicator is a second one)
        return struct.pack("<i", port + 3 * (CpuBand._ids - 1)) + bytes(124)

    def exchange_halo(self, name):
        has_up, has_down = self.rank > 0, self.rank < self.size - 1
        from_up, from_down = self._comm.exchange_rows(self.get_edge_row(name, 0) if has_up else None,
                                                      self.get_edge_row(name, 1) if has_down else None)
        return (self.set_halo_row(name, 0, from_up) if has_up else False, self.set_halo_row(name, 1, from_down) if has_down else False)

    def allreduce_max(self, value):
        return self._comm.allreduce_max(value) if self._comm is not None else value

    def _dtype(self, name):
        return {"dem": np.float32, "filled": np.float32, "depths": np.float32, "noflat": np.float64, "flowdir": np.uint8,
                "accum": np.float64, "labels": np.int32, "watersheds": np.int32, "ngdist": np.uint32}[name]

    def _raster(self, name):
        if name not in self.r:
            self.r[name] = np.zeros((self.H, self.W), dtype=self._dtype(name))
        return self.r[name]

    def upload(self, name, arr):
        self._raster(name)[self.ht:self.ht + self.nrows] = arr

    def download(self, name):
        return self.r[name][self.ht:self.ht + self.nrows].copy()

    def download_rows(self, name, row0, nrows):
        return self.r[name][self.ht + row0:self.ht + row0 + nrows].copy()

    def get_edge_row(self, name, side):
        row = {0: self.ht, 1: self.ht + self.nrows - 1, 2: 0, 3: self.H - 1}[side]
        return self._raster(name)[row].copy()

    def set_halo_row(self, name, side, row):
        a = self._raster(name)
        i = 0 if side == 0 else self.H - 1
        changed = not np.array_equal(a[i], row, equal_nan=True)
        a[i] = row
        return changed

    # device-buffer variants (HipBand.get_edge_row_dev / set_halo_row_dev): here "device" memory is host memory
    def row_bytes(self, name):
        return self.W * self._raster(name).dtype.itemsize

    def get_edge_row_dev(self, name, side, ptr):
        import ctypes
        row = np.ascontiguousarray(self.get_edge_row(name, side))
        ctypes.memmove(int(ptr), row.ctypes.data, row.nbytes)

    def set_halo_row_dev(self, name, side, ptr):
        import ctypes
        row = np.empty(self.W, dtype=self._raster(name).dtype)
        ctypes.memmove(row.ctypes.data, int(ptr), row.nbytes)
        return self.set_halo_row(name, side, row)

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

    def fill_attach(self, kind, short=0.0, diag=0.0):
        self.kind_state[kind] = (short, diag)
        self._attached = True
        return False

    def noflat_verify(self):
        return True

    def fill_halo_changed(self, kind, side):
        pass

    # ---- stand-in for the geodesic no-flats engine (csrc/noflat_geo.hip): same protocol, numpy arithmetic
    GEO_INF = 0xE0000000

    def geo_begin(self, short, diag):
        F = self.r["filled"].astype(np.float64)
        H, W = self.H, self.W
        gr = np.arange(H) + (self.row0 - self.ht)
        border = np.zeros((H, W), bool)
        border[(gr == 0) | (gr == self.Hg - 1), :] = True
        border[:, 0] = border[:, -1] = True
        owned = np.zeros((H, W), bool)
        owned[self.ht:self.ht + self.nrows] = True
        P = np.pad(F, 1, constant_values=np.nan)
        nb = [(-1, -1, 1), (-1, 0, 0), (-1, 1, 1), (0, -1, 0), (0, 1, 0), (1, -1, 1), (1, 0, 0), (1, 1, 1)]
        lower = np.zeros((H, W), bool)
        self._geo_same = []
        with np.errstate(invalid="ignore"):
            for dr, dc, dg in nb:
                N = P[1 + dr:1 + dr + H, 1 + dc:1 + dc + W]
                lower |= N < F
                self._geo_same.append((dr, dc, dg, N == F))
        flat = owned & ~border & ~lower
        up = np.nextafter(F, np.inf)
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            u = np.nextafter(up, np.inf) - up
            Su, Du = short / u, diag / u
        with np.errstate(invalid="ignore"):
            regular = np.isfinite(F) & (F != 0) & (np.rint(Su) == Su) & (Su >= 1) & (Du < 2 ** 28) & (np.abs(Du - np.rint(Du)) != 0.5)
        if np.isnan(self.r["filled"][owned]).any() or not (short > 0 and diag > 0):
            return False, False
        irregular = flat & ~regular       # left to the float64 relaxation (geo_end reports a partial surface)
        flat = flat & regular
        self._geo = dict(flat=flat, irregular=irregular, u=u, S=np.where(regular, np.rint(Su), 0).astype(np.int64), D=np.where(regular, np.rint(Du), 0).astype(np.int64),
                         short=short, diag=diag)
        d = self._raster("ngdist")
        d[...] = np.where(flat | irregular, self.GEO_INF, 0).astype(np.uint32)
        if self.ht:
            d[0] = self.GEO_INF       # the neighbour's cells: unknown until the first exchange
        if self.hb:
            d[-1] = self.GEO_INF
        self._geo_relax()
        return True, True

    def _geo_relax(self):
        g, H, W = self._geo, self.H, self.W
        INF = np.int64(self.GEO_INF)
        d = self.r["ngdist"].astype(np.int64)
        while True:
            Pd = np.pad(d, 1, constant_values=INF)
            best = d.copy()
            for dr, dc, dg, same in self._geo_same:
                N = Pd[1 + dr:1 + dr + H, 1 + dc:1 + dc + W]
                cand = np.where(same & g["flat"], np.minimum(N + (g["D"] if dg else g["S"]), INF), INF)
                best = np.minimum(best, cand)
            if np.array_equal(best, d):
                break
            d = best
        self.r["ngdist"][...] = d.astype(np.uint32)

    def geo_halo_changed(self, side):
        pass

    def geo_batch(self):
        self._geo_relax()
        return False

    def geo_end(self):
        F = self.r["filled"].astype(np.float64)
        d = self.r["ngdist"].astype(np.float64)
        up = np.nextafter(F, np.inf)
        with np.errstate(over="ignore", invalid="ignore"):
            u = np.nextafter(up, np.inf) - up
            G = np.where(d > 0, F + d * u, F)
        beyond = self.r["ngdist"] >= 0x80000000          # irregular flats, distances past the uint32 headroom
        short, diag = self._geo["short"], self._geo["diag"]
        G[beyond] = F[beyond] + 1.01 * self.Hg * self.W * diag
        self._raster("noflat")[...] = G
        self.kind_state[1] = (short, diag)
        return True, bool(beyond[self.ht:self.ht + self.nrows].any())

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

    def zero_raster(self, name):
        self._raster(name)[...] = 0

    def accum_boundary(self):
        """stand-in for mhip_ctx_band_accum_boundary: own contribution (halo rows = sources of no flux) + exit map"""
        fd = self.r["flowdir"]
        H, W = self.H, self.W
        DR = [-1, -1, 0, 1, 1, 1, 0, -1]
        DC = [0, 1, 1, 1, 0, -1, -1, -1]
        self.run_accum(halo_zero=True)
        halo = {0: 0} if self.ht else {}
        if self.hb:
            halo[H - 1] = 1
        exit_map = np.full(2 * W, -1, np.int32)
        for hr, side in halo.items():
            for c in range(W):
                r, cc, steps = hr, c, 0
                while steps <= H * W:
                    d = fd[r, cc]
                    if d > 7:
                        break
                    nr, nc = r + DR[d], cc + DC[d]
                    if not (0 <= nr < H and 0 <= nc < W):
                        break
                    if nr in halo:
                        if r not in halo:
                            exit_map[side * W + c] = halo[nr] * W + cc
                        break
                    r, cc, steps = nr, nc, steps + 1
        return exit_map

    def run_accum(self, halo_zero=False):
        """Kahn accumulation of the owned rows; halo cells are sources when their value is known (> 0), else they block."""
        fd, acc = self.r["flowdir"], self._raster("accum")
        H, W = self.H, self.W
        DR = [-1, -1, 0, 1, 1, 1, 0, -1]
        DC = [0, 1, 1, 1, 0, -1, -1, -1]
        halo = np.zeros(H, bool)
        if self.ht:
            halo[0] = True
        if self.hb:
            halo[H - 1] = True
        val = np.zeros((H, W))
        pend = np.zeros((H, W), int)
        for r in range(H):
            for c in range(W):
                if halo[r]:
                    val[r, c] = acc[r, c] if acc[r, c] > 0 and not halo_zero else 0
                    pend[r, c] = 0 if acc[r, c] > 0 or halo_zero else 99
                    continue
                val[r, c] = 1
                for k in range(8):
                    nr, nc = r + DR[k], c + DC[k]
                    if 0 <= nr < H and 0 <= nc < W and fd[nr, nc] <= 7 and fd[nr, nc] == (k + 4) % 8:
                        pend[r, c] += 1
        stack = [(r, c) for r in range(H) for c in range(W) if pend[r, c] == 0]
        done = np.zeros((H, W), bool)
        while stack:
            r, c = stack.pop()
            done[r, c] = True
            d = fd[r, c]
            if d > 7:
                continue
            nr, nc = r + DR[d], c + DC[d]
            if not (0 <= nr < H and 0 <= nc < W) or halo[nr]:
                continue
            val[nr, nc] += val[r, c]
            pend[nr, nc] -= 1
            if pend[nr, nc] == 0:
                stack.append((nr, nc))
        own = ~halo
        acc[own] = np.where(done[own], val[own], 0.0)

    def ccl_local(self):
        lab, n = oracle.connected_components(self.r["depths"])
        self._raster("labels")[...] = lab
        return n

    # the labelling in two halves, as the HIP band does it (mhip_ctx_band_ccl_begin / _finish): between the halves only the two top
    # and the two bottom rows of the labels raster hold (band-local) labels -- everything else is poisoned here, so a caller that
    # read more than the edge rows before the merge would be found out
    def ccl_begin(self):
        lab, n = oracle.connected_components(self.r["depths"])
        self._pending_labels = lab
        vis = self._raster("labels")
        vis[...] = -12345
        H = vis.shape[0]
        top = min(H, 2 if H >= 4 else H)
        vis[:top] = lab[:top]
        if H >= 4:
            vis[H - 2:] = lab[H - 2:]
        self._nlocal = n
        return n

    def ccl_finish(self, offset, dropped, target, nlabels_global, with_stats):
        self._raster("labels")[...] = self._pending_labels
        del self._pending_labels
        self.relabel_sparse(self._nlocal, offset, dropped, target, nlabels_global)
        if with_stats:
            self.records_compute(0)
            self._stats_with_labels = True

    def relabel(self, lut, nlabels_global):
        self.r["labels"][...] = np.asarray(lut)[self.r["labels"]]
        self.nlabels_global = int(nlabels_global)

    def relabel_sparse(self, nlocal, offset, dropped, target, nlabels_global):
        l = np.arange(nlocal + 1, dtype=np.int64)
        lut = offset + l - np.searchsorted(dropped, l)
        lut[np.asarray(dropped, dtype=np.int64)] = target
        lut[0] = 0
        self.relabel(lut, nlabels_global)

    def relabel_range(self, lo, hi, lut, foreign_ids, foreign_new, nlabels_new):
        lab = self.r["labels"]
        out = np.zeros_like(lab)
        own = (lab >= lo) & (lab <= hi)
        out[own] = np.asarray(lut, dtype=np.int64)[lab[own] - lo]
        other = (lab > 0) & ~own
        fid, fnew = np.asarray(foreign_ids, dtype=np.int64), np.asarray(foreign_new, dtype=np.int64)
        pos = np.searchsorted(fid, lab[other])
        ok = (pos < fid.size) & (fid[np.minimum(pos, max(fid.size - 1, 0))] == lab[other]) if fid.size else np.zeros(int(other.sum()), bool)
        out[other] = np.where(ok, fnew[np.minimum(pos, max(fid.size - 1, 0))] if fid.size else 0, 0)
        lab[...] = out
        self.nlabels_global = int(nlabels_new)

    def trace(self, cells, src, background_label, geometry):
        fd, lab = self.r["flowdir"], self.r["labels"]
        DR = [-1, -1, 0, 1, 1, 1, 0, -1]
        DC = [0, 1, 1, 1, 0, -1, -1, -1]
        row_lo, own0, own1 = self.row0 - self.ht, self.ht, self.ht + self.nrows
        n = len(cells)
        out_lab, status, src_out = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        exits, geoms = np.full((n, 2), -1, np.int64), []
        for i in range(n):
            gr, c = int(cells[i][0]), int(cells[i][1])
            r = gr - row_lo
            s_ = int(src[i])
            g = []
            if 0 <= gr < self.Hg and 0 <= c < self.W and own0 <= r < own1:
                if s_ < 0:
                    s_ = int(lab[r, c])
                for _ in range(self.Hg * self.W):
                    g.append((r + row_lo) * self.W + c)
                    l = int(lab[r, c])
                    if l != s_ and (background_label is None or l != background_label):
                        status[i], out_lab[i] = 1, l
                        break
                    k = int(fd[r, c])
                    if k > 7:
                        break
                    r, c = r + DR[k], c + DC[k]
                    if not (0 <= r + row_lo < self.Hg and 0 <= c < self.W):
                        break
                    if not (own0 <= r < own1):
                        status[i] = 2
                        exits[i] = (r + row_lo, c)
                        break
            src_out[i] = s_
            geoms.append(np.asarray(g, dtype=np.int64))
        return out_lab, status, src_out, exits, (geoms if geometry else [None] * n)

    def watershed_local(self):
        ws = self._raster("watersheds")
        ws[...] = self.r["labels"]
        if self.ht:
            ws[0] = -(1 + np.arange(self.W))
        if self.hb:
            ws[-1] = -(1 + self.W + np.arange(self.W))
        # nearest labelled cell downstream (labels incl. the negative pseudo labels are terminals), 0 when the path leaves
        fd = self.r["flowdir"]
        DR = [-1, -1, 0, 1, 1, 1, 0, -1]
        DC = [0, 1, 1, 1, 0, -1, -1, -1]
        out = ws.copy()
        for r in range(self.H):
            for c in range(self.W):
                if ws[r, c] != 0:
                    continue
                pr, pc, res = r, c, 0
                for _ in range(self.H * self.W):
                    d = fd[pr, pc]
                    if d > 7:
                        break
                    pr, pc = pr + DR[d], pc + DC[d]
                    if not (0 <= pr < self.H and 0 <= pc < self.W):
                        break
                    if ws[pr, pc] != 0:
                        res = ws[pr, pc]
                        break
                out[r, c] = res
        ws[...] = out

    def apply_neg_lut(self, name, lut):
        a = self.r[name]
        neg = a < 0
        a[neg] = np.asarray(lut)[-a[neg] - 1]

    def get_int(self, key):
        return 0

    def _owned(self, name):
        return self.r[name][self.ht:self.ht + self.nrows]

    # device-resident record sets of HipBand, here plain arrays: which = 0 stats, 1 watershed counts, 2 pour points
    def records_compute(self, which):
        n = self.nlabels_global
        if which == 0:
            rec = oracle.label_stats(self._owned("depths"), self._owned("labels"), n)
        elif which == 1:
            rec = np.bincount(self._owned("watersheds").ravel(), minlength=n + 1).astype(np.int64)
        else:
            rec = (oracle.label_max_index(self._owned("accum"), self._owned("labels"), n) if which == 2 else
                   oracle.label_min_index(self._owned("noflat"), self._owned("labels"), n))
            rec["row"] = np.where(rec["row"] >= 0, rec["row"] + self.row0, rec["row"])
        self._rec = getattr(self, "_rec", {})
        self._rec[which] = rec

    def records_fetch(self, which, first, count):
        return self._rec[which][first:first + count].copy()

    def records_gather(self, which, ids):
        return self._rec[which][np.asarray(ids, dtype=np.int64)].copy()

    def foreign_counts(self, lo, hi):
        cnt = self._rec[1]
        ids = np.flatnonzero(cnt)
        ids = ids[(ids != 0) & ((ids < lo) | (ids > hi))]
        return ids.astype(np.int64), cnt[ids]
