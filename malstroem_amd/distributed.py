"""Row-band decomposition of one large DEM over several GPUs.

Each rank owns a contiguous band of rows plus one halo row per neighbour.  The fills (plain and no-flats) are
global fixed points: every band iterates locally (``mhip_ctx_fill_batch``), neighbours swap their edge rows, a
band whose halo changed re-activates the tiles next to it, and an all-reduce of "somebody is still active"
ends the loop -- the result is bit-identical to the single-raster fill because the iteration is monotone and
schedule independent (csrc/fill.hip).  D8 needs one halo row of the no-flats surface and hands its own edge
rows to the neighbours afterwards (accumulation / watersheds walk across bands on them).

Transport.  The DATA path -- neighbour exchange of edge rows and the one-word "anybody still active" reduction -- runs
over RCCL inside the library (``mhip_ctx_exchange_halo`` / ``mhip_ctx_allreduce_max``: ncclSend/ncclRecv between band
neighbours over xGMI on the band's own stream, no host hop).  The CONTROL path -- handing the 128-byte ncclUniqueId to
every rank and the few small Python-object collectives of the label / watershed boundary systems -- goes through a
``Comm`` the launcher supplies: ``SocketComm`` (stdlib TCP, rank 0 as hub: nothing but Python is needed),
``ThreadComm`` (several bands inside one process: tests on a single GPU), or any object with the same four methods
(``tools/launch_comm.TorchComm`` wraps torch.distributed/gloo for launchers that already live in that world; this
package itself never imports it).  Without RCCL (one GPU shared by several bands, CPU stand-in backends in the
tests) the rows travel through the ``Comm`` as host buffers.  The compute backend is ``HipBand`` (the C-ABI band context).
"""
import ctypes
import os
import pickle
import queue
import socket
import struct
import sys
import threading

import numpy as np

from . import _lib
from ._lib import INDEX_DTYPE, R_DEM, R_FILLED, R_FLOWDIR, R_NOFLAT, RASTER_DTYPE, STAGE_ACCUM, STAGE_FLOWDIR, STAT_DTYPE
from .pipeline import RASTERS

__all__ = ["band_rows", "Comm", "SingleComm", "ThreadComm", "SocketComm", "HybridComm", "HipBand", "BandPipeline"]


TILE = 62     # tile edge of the fill kernels (csrc/pflood.hip, csrc/noflat_geo.hip): window 64 = tile + ring


def band_rows(H, size, rank, align=True):
    """(row0, nrows) of band ``rank``: contiguous row bands of near-equal height.

    ``align``: the seams between bands are put on the tile grid of the fill kernels (a band's last owned row is the last row
    of a tile: global row 62 k), so that the tiles of the bands are the tiles of the undivided raster and a band's bottom halo
    row is a window ring row -- what the tiled priority-flood needs to run on a band (otherwise the band falls back to the
    iterative tile schedule).  Needs at least one tile row per band; the heights then differ by at most one tile row."""
    H, size, rank = int(H), int(size), int(rank)
    if H < size:
        raise ValueError("more bands (%d) than raster rows (%d)" % (size, H))
    ntr = -(-(H - 2) // TILE) if H > 2 else 0      # tile rows of the undivided raster (its border rows belong to no tile)
    if align and size > 1 and ntr >= size:
        # seam b (first row of band b) = the tile-grid row 1 + 62 k nearest to the equal split, kept strictly increasing
        seams = [0]
        for b in range(1, size):
            k = int(round((b * H / float(size) - 1) / TILE))
            k = max(k, (seams[-1] - 1) // TILE + 1 if seams[-1] else 1)      # at least one tile row for the band above ...
            k = min(k, ntr - (size - b))                                     # ... and for every band below
            seams.append(1 + TILE * k)
        seams.append(H)
        if all(seams[i] < seams[i + 1] for i in range(size)):
            return seams[rank], seams[rank + 1] - seams[rank]
    base, extra = divmod(H, size)
    row0 = rank * base + min(rank, extra)
    return row0, base + (1 if rank < extra else 0)


# ---- transports -----------------------------------------------------------------------------------------------

class Comm(object):
    """Minimal neighbour/collective interface the band protocol needs."""
    rank = 0
    size = 1

    def exchange_rows(self, to_up, to_down):
        """Send ``to_up`` to rank-1 and ``to_down`` to rank+1 (None at the raster ends); returns (from_up, from_down)."""
        raise NotImplementedError

    def allreduce_max(self, value):
        raise NotImplementedError

    def allgather(self, obj):
        raise NotImplementedError

    def clone(self):
        """A second, independent communicator over the same ranks (collective call): the labelling branch of
        BandPipeline.run_chain talks on it while the main thread keeps using this one."""
        raise NotImplementedError


class SingleComm(Comm):
    def clone(self):
        return SingleComm()

    def exchange_rows(self, to_up, to_down):
        return None, None

    def allreduce_max(self, value):
        return value

    def allgather(self, obj):
        return [obj]


class ThreadComm(Comm):
    """In-process transport: ``ThreadComm.world(n)`` returns n endpoints to be driven by n threads."""

    class _World(object):
        def __init__(self, n):
            self.n = n
            self.down = [queue.Queue() for _ in range(n)]   # down[i]: messages travelling from rank i to rank i+1
            self.up = [queue.Queue() for _ in range(n)]     # up[i]: messages travelling from rank i to rank i-1
            self.barrier = threading.Barrier(n)
            self.slots = [None] * n

    @classmethod
    def world(cls, n):
        w = cls._World(n)
        return [cls(w, r) for r in range(n)]

    def __init__(self, world, rank):
        self._w = world
        self.rank = rank
        self.size = world.n

    def exchange_rows(self, to_up, to_down):
        w = self._w
        if self.rank > 0:
            w.up[self.rank].put(None if to_up is None else np.array(to_up, copy=True))
        if self.rank < self.size - 1:
            w.down[self.rank].put(None if to_down is None else np.array(to_down, copy=True))
        from_up = w.down[self.rank - 1].get(timeout=600) if self.rank > 0 else None
        from_down = w.up[self.rank + 1].get(timeout=600) if self.rank < self.size - 1 else None
        return from_up, from_down

    def allgather(self, obj):
        w = self._w
        w.slots[self.rank] = obj
        w.barrier.wait(timeout=600)
        out = list(w.slots)
        w.barrier.wait(timeout=600)
        return out

    def allreduce_max(self, value):
        return max(self.allgather(value))

    def clone(self):
        w = self.allgather(ThreadComm._World(self.size) if self.rank == 0 else None)[0]   # rank 0's object, shared in-process
        return ThreadComm(w, self.rank)


class SocketComm(Comm):
    """Control-plane transport over plain TCP sockets (stdlib only): rank 0 listens on (addr, port), every other rank
    connects once; a collective is "send my pickled object to rank 0, receive the list of everybody's".  Messages are a
    few rows at most (the boundary systems of the label / watershed protocols, the 128-byte ncclUniqueId), so the star
    topology is not a bottleneck; the rows of the fills travel over RCCL, not through here.

    ``SocketComm.from_env()`` reads RANK / WORLD_SIZE / MASTER_ADDR / MALSTROEM_COMM_PORT (default MASTER_PORT + 17)."""

    one_rank_per_device = True   # ranks are processes, by convention one per GPU: BandPipeline may put the rows on RCCL

    def __init__(self, rank, size, addr="127.0.0.1", port=29517, timeout_s=300.0, _socks=None):
        self.rank, self.size, self._addr, self._port, self._timeout = int(rank), int(size), addr, int(port), float(timeout_s)
        self._lock = threading.Lock()
        self._nclones = 0
        if _socks is not None:
            self._peers, self._hub = _socks
            return
        self._peers, self._hub = {}, None
        if self.size == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, self._port))
            srv.listen(self.size)
            srv.settimeout(self._timeout)
            try:
                while len(self._peers) < self.size - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(self._timeout)
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    r = struct.unpack("<i", self._recv_exact(conn, 4))[0]
                    self._peers[r] = conn
            finally:
                srv.close()
        else:
            import time
            deadline = time.time() + self._timeout
            while True:
                try:
                    self._hub = socket.create_connection((addr, self._port), timeout=self._timeout)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            self._hub.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._hub.sendall(struct.pack("<i", self.rank))

    @classmethod
    def from_env(cls, env=None):
        import os
        env = os.environ if env is None else env
        port = int(env.get("MALSTROEM_COMM_PORT", int(env.get("MASTER_PORT", "29500")) + 17))
        return cls(int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), env.get("MASTER_ADDR", "127.0.0.1"), port)

    @staticmethod
    def _recv_exact(sock, n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("SocketComm: peer closed the connection")
            buf += chunk
        return bytes(buf)

    @classmethod
    def _send_msg(cls, sock, payload):
        sock.sendall(struct.pack("<q", len(payload)) + payload)

    @classmethod
    def _recv_msg(cls, sock):
        n = struct.unpack("<q", cls._recv_exact(sock, 8))[0]
        return cls._recv_exact(sock, n)

    def allgather(self, obj):
        if self.size == 1:
            return [obj]
        with self._lock:
            if self.rank == 0:
                parts = [None] * self.size
                parts[0] = obj
                for r, conn in self._peers.items():
                    parts[r] = pickle.loads(self._recv_msg(conn))
                blob = pickle.dumps(parts, protocol=pickle.HIGHEST_PROTOCOL)
                for conn in self._peers.values():
                    self._send_msg(conn, blob)
                return parts
            self._send_msg(self._hub, pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL))
            return pickle.loads(self._recv_msg(self._hub))

    def allreduce_max(self, value):
        return max(self.allgather(float(value)))

    def exchange_rows(self, to_up, to_down):
        """host-staged neighbour exchange (bands without RCCL): everybody's edge rows pass through the hub"""
        rows = self.allgather((to_up, to_down))
        from_up = rows[self.rank - 1][1] if self.rank > 0 else None
        from_down = rows[self.rank + 1][0] if self.rank < self.size - 1 else None
        return from_up, from_down

    def clone(self):
        """a second, independent set of connections on the next port (collective call)"""
        self._nclones += 1
        return SocketComm(self.rank, self.size, self._addr, self._port + self._nclones, self._timeout)

    def close(self):
        for c in list(self._peers.values()) + ([self._hub] if self._hub else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._hub = {}, None


class HybridComm(Comm):
    """``k`` bands per process (threads) times ``P`` processes = ``P * k`` virtual ranks, band order = (process, thread).

    A GPU addresses at most 2**31 - 2 cells per band context (int32 cell indices keep the union-find / pointer-jumping
    rasters at 4 bytes per cell), so a raster beyond that runs as several bands on ONE GPU as well: the 1- and 2-GPU
    points of a 65536 x 65536 strong-scaling curve are 4 and 2 bands per process.  Several bands share a GPU here, so
    there is no RCCL communicator (one rank per device); rows and objects travel through the process-level ``Comm``
    (all-gathered: the few boundary rows of every band), which is fine for what this mode is for.

    ``HybridComm.world(proc_comm, k)`` -> the ``k`` endpoints of this process, one per band thread."""

    class _World(object):
        def __init__(self, proc, k):
            self.proc, self.k = proc, k
            self.barrier = threading.Barrier(k)
            self.slots = [None] * k
            self.result = None

    @classmethod
    def world(cls, proc_comm, k):
        w = cls._World(proc_comm, int(k))
        return [cls(w, t) for t in range(int(k))]

    def __init__(self, world, t):
        self._w, self._t = world, t
        self.rank = world.proc.rank * world.k + t
        self.size = world.proc.size * world.k

    def allgather(self, obj):
        w = self._w
        w.slots[self._t] = obj
        w.barrier.wait(timeout=1800)
        if self._t == 0:
            w.result = [o for part in w.proc.allgather(list(w.slots)) for o in part]
        w.barrier.wait(timeout=1800)
        out = w.result
        w.barrier.wait(timeout=1800)
        return out

    def allreduce_max(self, value):
        return max(self.allgather(float(value)))

    def exchange_rows(self, to_up, to_down):
        rows = self.allgather((to_up, to_down))
        from_up = rows[self.rank - 1][1] if self.rank > 0 else None
        from_down = rows[self.rank + 1][0] if self.rank < self.size - 1 else None
        return from_up, from_down

    def clone(self):
        w = self._w
        w.barrier.wait(timeout=1800)
        if self._t == 0:
            w.result = HybridComm._World(w.proc.clone(), w.k)
        w.barrier.wait(timeout=1800)
        nw = w.result
        w.barrier.wait(timeout=1800)
        return HybridComm(nw, self._t)


# ---- compute backend: one band context on one GPU -----------------------------------------------------------------

class HipBand(object):
    """ctypes face of a band ``mhip_ctx`` (include/malstroem_hip.h, row-band protocol)."""

    def __init__(self, H_global, W, row0, nrows, device=0, rank=0, size=1, unique_id=None):
        """``unique_id``: the 128 bytes of rank 0's ``HipBand.new_unique_id()``; with it the context joins the RCCL
        communicator of all ``size`` bands (a collective call) and moves its halo rows itself."""
        self.W, self.nrows = int(W), int(nrows)
        self._ctx = ctypes.c_void_p()
        uid = None
        if unique_id is not None:
            if len(unique_id) != 128:
                raise ValueError("ncclUniqueId must be 128 bytes")
            uid = ctypes.create_string_buffer(bytes(unique_id), 128)
        _lib.call("mhip_ctx_create_band", ctypes.byref(self._ctx), _lib.i64(H_global), _lib.i64(W), _lib.i64(row0),
                  _lib.i64(nrows), int(device), int(rank), int(size), uid)
        self.has_comm = bool(_lib.load().mhip_ctx_has_comm(self._ctx))

    @staticmethod
    def new_unique_id():
        """ncclGetUniqueId through the library (rank 0 calls it, the launcher's ``Comm`` distributes the bytes)."""
        buf = ctypes.create_string_buffer(128)
        _lib.call("mhip_comm_unique_id", buf)
        return bytes(buf.raw)

    def exchange_halo(self, name):
        """RCCL neighbour exchange of raster ``name``'s edge rows -> (top halo changed, bottom halo changed)."""
        changed = (ctypes.c_int32 * 2)(0, 0)
        _lib.call("mhip_ctx_exchange_halo", self._ctx, RASTERS[name], changed)
        return bool(changed[0]), bool(changed[1])

    def allreduce_max(self, value):
        out = ctypes.c_double(0.0)
        _lib.call("mhip_ctx_allreduce_max", self._ctx, ctypes.c_double(float(value)), ctypes.byref(out))
        return out.value

    def close(self):
        if self._ctx:
            _lib.call("mhip_ctx_destroy", self._ctx)
            self._ctx = ctypes.c_void_p()

    def upload(self, name, arr):
        which = RASTERS[name]
        a = np.ascontiguousarray(arr, dtype=RASTER_DTYPE[which])
        if a.shape != (self.nrows, self.W):
            raise ValueError("band raster must be %s, got %s" % ((self.nrows, self.W), a.shape))
        _lib.call("mhip_ctx_upload", self._ctx, which, _lib.ptr(a))

    def download(self, name):
        which = RASTERS[name]
        out = np.empty((self.nrows, self.W), dtype=RASTER_DTYPE[which])
        _lib.call("mhip_ctx_download", self._ctx, which, _lib.ptr(out))
        return out

    def get_edge_row(self, name, side):
        which = RASTERS[name]
        out = np.empty(self.W, dtype=RASTER_DTYPE[which])
        _lib.call("mhip_ctx_get_edge_row", self._ctx, which, int(side), _lib.ptr(out))
        return out

    def set_halo_row(self, name, side, row):
        which = RASTERS[name]
        a = np.ascontiguousarray(row, dtype=RASTER_DTYPE[which])
        changed = ctypes.c_int32(0)
        _lib.call("mhip_ctx_set_halo_row", self._ctx, which, int(side), _lib.ptr(a), ctypes.byref(changed))
        return bool(changed.value)

    def row_bytes(self, name):
        return self.W * np.dtype(RASTER_DTYPE[RASTERS[name]]).itemsize

    def get_edge_row_dev(self, name, side, dev_ptr):
        """Copy an edge row into a DEVICE buffer of the caller (W * itemsize bytes on this band's GPU)."""
        _lib.call("mhip_ctx_get_edge_row_dev", self._ctx, RASTERS[name], int(side), ctypes.c_void_p(int(dev_ptr)))

    def set_halo_row_dev(self, name, side, dev_ptr):
        changed = ctypes.c_int32(0)
        _lib.call("mhip_ctx_set_halo_row_dev", self._ctx, RASTERS[name], int(side), ctypes.c_void_p(int(dev_ptr)),
                  ctypes.byref(changed))
        return bool(changed.value)

    def dem_minmax(self):
        mn, mx, nan = ctypes.c_float(0), ctypes.c_float(0), ctypes.c_int32(0)
        _lib.call("mhip_ctx_dem_minmax", self._ctx, ctypes.byref(mn), ctypes.byref(mx), ctypes.byref(nan))
        return np.float32(mn.value), np.float32(mx.value), bool(nan.value)

    def fill_begin(self, kind, short=0.0, diag=0.0):
        active = ctypes.c_int32(0)
        _lib.call("mhip_ctx_fill_begin", self._ctx, int(kind), ctypes.c_double(short), ctypes.c_double(diag), ctypes.byref(active))
        return bool(active.value)

    def fill_batch(self, kind):
        active = ctypes.c_int32(0)
        _lib.call("mhip_ctx_fill_batch", self._ctx, int(kind), ctypes.byref(active))
        return bool(active.value)

    def fill_halo_changed(self, kind, side):
        _lib.call("mhip_ctx_fill_halo_changed", self._ctx, int(kind), int(side))

    def fill_certify(self, kind):
        """one sweep over every tile of the band, iterated to local convergence -> did any tile move?"""
        changed = ctypes.c_int32(0)
        _lib.call("mhip_ctx_fill_certify", self._ctx, int(kind), ctypes.byref(changed))
        return bool(changed.value)

    def fill_end(self, kind):
        _lib.call("mhip_ctx_fill_end", self._ctx, int(kind))

    # the no-flats fill as an integer geodesic distance transform (csrc/noflat_geo.hip); edge rows of raster "ngdist" travel
    def geo_begin(self, short, diag):
        """-> (applicable, active).  Not applicable (a flat at elevation 0, NaN, ...): run the float64 relaxation (kind 1)."""
        ap, ac = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.call("mhip_ctx_geo_begin", self._ctx, ctypes.c_double(short), ctypes.c_double(diag), ctypes.byref(ap), ctypes.byref(ac))
        return bool(ap.value), bool(ac.value)

    def geo_batch(self):
        active = ctypes.c_int32(0)
        _lib.call("mhip_ctx_geo_batch", self._ctx, ctypes.byref(active))
        return bool(active.value)

    def geo_halo_changed(self, side):
        _lib.call("mhip_ctx_geo_halo_changed", self._ctx, int(side))

    def geo_end(self):
        """writes the no-flats surface -> (ok, partial).  Not partial: the reference's equation held at every owned cell (ok).
        Partial: exact except on flats the transform does not cover, which hold an upper bound for the relaxation."""
        ok, partial = ctypes.c_int32(0), ctypes.c_int32(0)
        _lib.call("mhip_ctx_geo_end", self._ctx, ctypes.byref(ok), ctypes.byref(partial))
        return bool(ok.value), bool(partial.value)

    def fill_attach(self, kind, short=0.0, diag=0.0):
        """fill_begin without the initialising round: the raster already holds an upper bound of the fixed point"""
        _lib.call("mhip_ctx_fill_attach", self._ctx, int(kind), ctypes.c_double(short), ctypes.c_double(diag))
        return False

    def noflat_verify(self):
        ok = ctypes.c_int32(0)
        _lib.call("mhip_ctx_noflat_verify", self._ctx, ctypes.byref(ok))
        return bool(ok.value)

    def run_flowdir(self):
        _lib.call("mhip_ctx_run", self._ctx, STAGE_FLOWDIR)
        _lib.call("mhip_ctx_sync", self._ctx)

    def zero_raster(self, name):
        _lib.call("mhip_ctx_zero_raster", self._ctx, RASTERS[name])

    def run_accum(self):
        _lib.call("mhip_ctx_run", self._ctx, STAGE_ACCUM)
        _lib.call("mhip_ctx_sync", self._ctx)

    def accum_boundary(self):
        """Boundary pass of the band accumulation (``mhip_ctx_band_accum_boundary``): ACCUM = the band's own contribution;
        returns exit_map[2 * W] (halo cell -> side * W + column of the edge cell its flux leaves the band from, or -1)."""
        out = np.empty(2 * self.W, dtype=np.int32)
        _lib.call("mhip_ctx_band_accum_boundary", self._ctx, _lib.ptr(out))
        return out

    def ccl_local(self):
        n = ctypes.c_int64(0)
        _lib.call("mhip_ctx_band_ccl_local", self._ctx, ctypes.byref(n))
        return int(n.value)

    def relabel(self, lut, nlabels_global):
        lut = np.ascontiguousarray(lut, dtype=np.int32)
        _lib.call("mhip_ctx_band_relabel", self._ctx, _lib.ptr(lut), _lib.i64(lut.size - 1), _lib.i64(nlabels_global))

    def relabel_sparse(self, nlocal, offset, dropped, target, nlabels_global):
        """local label l -> offset + l - #(dropped < l); dropped[k] (sorted) -> target[k]"""
        d = np.ascontiguousarray(dropped, dtype=np.int32)
        t = np.ascontiguousarray(target, dtype=np.int32)
        _lib.call("mhip_ctx_band_relabel_sparse", self._ctx, _lib.i64(nlocal), _lib.i64(offset), _lib.ptr(d), _lib.ptr(t), _lib.i64(d.size),
                  _lib.i64(nlabels_global))

    def watershed_local(self):
        _lib.call("mhip_ctx_band_watershed_local", self._ctx)

    def apply_neg_lut(self, name, lut):
        lut = np.ascontiguousarray(lut, dtype=np.int32)
        _lib.call("mhip_ctx_band_apply_neg_lut", self._ctx, RASTERS[name], _lib.ptr(lut), _lib.i64(lut.size))

    def get_int(self, key):
        v = ctypes.c_int64(0)
        _lib.call("mhip_ctx_get_i64", self._ctx, key.encode(), ctypes.byref(v))
        return v.value

    def side_begin(self):
        """The calling thread's band calls run on the context's side stream from here on (after everything issued so far)."""
        _lib.call("mhip_ctx_side_begin", self._ctx)

    def side_end(self):
        _lib.call("mhip_ctx_side_end", self._ctx)

    # per-label records over the OWNED rows, indexed by global label; they stay on the device, the launcher fetches pieces
    _REC_DTYPE = (STAT_DTYPE, np.dtype(np.int64), INDEX_DTYPE)      # which = 0 stats, 1 watershed counts, 2 pour points

    def records_compute(self, which):
        _lib.call("mhip_ctx_band_records", self._ctx, int(which))

    def records_fetch(self, which, first, count):
        out = np.zeros(int(count), dtype=self._REC_DTYPE[which])
        _lib.call("mhip_ctx_band_fetch", self._ctx, int(which), _lib.i64(first), _lib.i64(count), _lib.ptr(out))
        return out

    def records_gather(self, which, ids):
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        out = np.zeros(ids.size, dtype=self._REC_DTYPE[which])
        _lib.call("mhip_ctx_band_gather", self._ctx, int(which), _lib.ptr(ids), _lib.i64(ids.size), _lib.ptr(out))
        return out

    def foreign_counts(self, lo, hi):
        cap = 1 << 16
        while True:
            ids, vals, n = np.zeros(cap, np.int64), np.zeros(cap, np.int64), ctypes.c_int64(0)
            _lib.call("mhip_ctx_band_foreign_counts", self._ctx, _lib.i64(lo), _lib.i64(hi), _lib.i64(cap), _lib.ptr(ids), _lib.ptr(vals),
                      ctypes.byref(n))
            if n.value <= cap:
                order = np.argsort(ids[:n.value])
                return ids[:n.value][order], vals[:n.value][order]
            cap = int(n.value)


# ---- the protocol ---------------------------------------------------------------------------------------------------

def solve_band_accum(info, W):
    """The seam system of the band accumulation.  ``info[b]`` = dict(a0=(first owned row, last owned row) of band b's own
    contribution, exit=exit_map of ``mhip_ctx_band_accum_boundary``).  Node (b, s, c) = cell c of band b's first (s = 0) /
    last (s = 1) row AS SEEN BY the neighbour on that side.  exit[k] = e for the halo cell k of band b makes the neighbour's
    node under k a child of node (b, e): its whole flux arrives at e.  Leaves first (Kahn), vectorised per level; a node
    whose own contribution is 0 (a flow cycle upstream, _flow.pyx:212-247 leaves such cells 0) or that has an unknown child
    stays unknown = 0.  Returns per band (top halo row, bottom halo row) of known accumulations, 0 = unknown."""
    R = len(info)
    N = R * 2 * W
    val = np.concatenate([np.asarray(r, dtype=np.float64) for i in info for r in i["a0"]])
    parent = np.full(N, -1, dtype=np.int64)
    cols = np.arange(W)
    for b in range(R):
        ex = np.asarray(info[b]["exit"], dtype=np.int64)
        if b > 0:           # my top halo row = the last row of band b - 1
            m = ex[:W] >= 0
            parent[((b - 1) * 2 + 1) * W + cols[m]] = b * 2 * W + ex[:W][m]
        if b < R - 1:       # my bottom halo row = the first row of band b + 1
            m = ex[W:] >= 0
            parent[((b + 1) * 2) * W + cols[m]] = b * 2 * W + ex[W:][m]
    # Kahn's walk over the forest, one O(N) pass in the library (host C++; as NumPy levels -- ufunc.at, unique -- it cost more
    # than the band's GPU work at 8 bands of 65536 columns)
    val = np.ascontiguousarray(val)
    _lib.call("mhip_band_forest_solve", _lib.i64(N), _lib.ptr(parent), _lib.ptr(val))
    val = val.reshape(R, 2, W)
    return [(val[b - 1, 1] if b > 0 else None, val[b + 1, 0] if b < R - 1 else None) for b in range(R)]


_PROF = os.environ.get("MALSTROEM_BAND_PROFILE") is not None    # development: wall clock of the steps inside label() / _merged()


class _Lap:
    """with-less lap timer: lap(name) charges the time since the previous lap to `name`; report() prints (rank 0 only)"""
    def __init__(self, what, rank):
        import time
        self.t, self.what, self.rank, self.acc, self.clock = time.perf_counter(), what, rank, [], time.perf_counter

    def lap(self, name):
        if _PROF:
            now = self.clock()
            self.acc.append((name, (now - self.t) * 1e3))
            self.t = now

    def report(self):
        if _PROF and self.rank == 0:
            print("[band profile] %s: %s" % (self.what, ", ".join("%s %.1f" % kv for kv in self.acc)), file=sys.stderr, flush=True)


class BandPipeline(object):
    """One rank's share of a row-banded DEM.  Every rank calls the same methods in the same order (SPMD)."""

    def __init__(self, comm, shape, device=0, backend_factory=None, rccl=None, align=True):
        """``align``: seams on the tile grid of the fill kernels (see ``band_rows``).  ``rccl``: True = the bands join an RCCL communicator inside the library and move their halo rows GPU -> GPU
        (one rank per GPU required), False = rows travel through ``comm`` as host buffers, None = RCCL whenever there is
        more than one rank and the backend is the HIP one.  Creation is voted on: if any rank fails, every rank raises."""
        self.comm = comm
        self.H, self.W = int(shape[0]), int(shape[1])
        self.row0, self.nrows = band_rows(self.H, comm.size, comm.rank, align)
        factory = backend_factory or HipBand
        if rccl is None:   # RCCL wants one rank per device: only transports between processes say so (SocketComm; a launcher's own)
            rccl = comm.size > 1 and factory is HipBand and bool(getattr(comm, "one_rank_per_device", False))
        self.rccl_error = None
        for attempt in (0, 1):
            self.band, err = None, None
            try:
                uid = None
                if rccl and comm.size > 1:
                    try:
                        mine = getattr(factory, "new_unique_id", HipBand.new_unique_id)() if comm.rank == 0 else None
                    except Exception as e:
                        mine, err = None, e
                    uid = comm.allgather(mine)[0]          # every rank takes part, whatever happened on rank 0
                    if uid is None:
                        raise err or RuntimeError("rank 0 could not create an ncclUniqueId")
                kw = dict(unique_id=uid) if uid is not None else {}
                self.band = factory(self.H, self.W, self.row0, self.nrows, device=device, rank=comm.rank, size=comm.size, **kw)
                if uid is not None:                          # one collective before anything depends on the communicator
                    self.band.allreduce_max(float(comm.rank))
            except Exception as e:
                err = err or e
            if comm.allreduce_max(1.0 if err is not None else 0.0) == 0.0:
                break
            if self.band is not None:
                self.band.close()
                self.band = None
            if not (rccl and comm.size > 1 and attempt == 0):
                raise err if err is not None else RuntimeError("band setup failed on another rank")
            # the RCCL communicator could not be set up on some rank: the same bands with their rows through `comm` (host buffers).
            # Every rank takes this branch together (the vote above); the reason is kept for the caller to report.
            self.rccl_error = repr(err) if err is not None else "RCCL setup failed on another rank"
            rccl = False
        self.rccl = bool(getattr(self.band, "has_comm", False))
        self.has_up = comm.rank > 0
        self.has_down = comm.rank < comm.size - 1
        self.exchanges = {"fill": 0, "noflat": 0}
        self.short = self.diag = None
        self._tls = threading.local()

    def close(self):
        self.band.close()

    # ---- helpers
    def _allreduce_max(self, value):
        """data-path reduction of the iteration loops: RCCL when the band owns a communicator"""
        return self.band.allreduce_max(value) if self.rccl else self.comm.allreduce_max(value)

    def _vote(self, busy, err=None):
        """all-reduce of "still busy" that also carries failures: a rank that caught an exception votes 2 and every
        rank leaves the loop with an error instead of waiting for the failed one in the next collective"""
        m = self._allreduce_max(2.0 if err is not None else (1.0 if busy else 0.0))
        if m >= 2.0:
            raise err if err is not None else RuntimeError("another band failed in this stage")
        return m > 0.0

    def _swap_edges(self, name):
        """Neighbours trade edge rows of raster ``name``; returns which of my halo rows changed (top, bottom)."""
        b = self.band
        if self.rccl:
            return b.exchange_halo(name)
        to_up = b.get_edge_row(name, 0) if self.has_up else None
        to_down = b.get_edge_row(name, 1) if self.has_down else None
        from_up, from_down = self.comm.exchange_rows(to_up, to_down)
        ch_top = b.set_halo_row(name, 0, from_up) if self.has_up else False
        ch_bot = b.set_halo_row(name, 1, from_down) if self.has_down else False
        return ch_top, ch_bot

    def upload_dem(self, dem_band):
        """``dem_band``: this rank's owned rows (nrows x W float32).  DEM halo rows are fetched from the neighbours."""
        self.band.upload("dem", dem_band)
        self._swap_edges("dem")

    def _fill(self, kind, name, short=0.0, diag=0.0, attach=False):
        b = self.band
        err, active = None, False
        try:
            active = b.fill_attach(kind, short, diag) if attach else b.fill_begin(kind, short, diag)
        except Exception as e:
            err = e
        while True:
            busy = False
            if err is None:
                try:
                    ch_top, ch_bot = self._swap_edges(name)
                    if ch_top:
                        b.fill_halo_changed(kind, 0)
                    if ch_bot:
                        b.fill_halo_changed(kind, 1)
                    self.exchanges["noflat" if kind else "fill"] += 1
                    busy = active or ch_top or ch_bot
                except Exception as e:
                    err = e
            if not self._vote(busy, err):
                # every band is quiescent: certify -- one sweep over EVERY tile of every band.  The worklist schedule only
                # revisits a tile when a neighbour's probe saw its halo drop; a sweep that moves nothing anywhere proves the
                # state is the global fixed point (and repairs a lost wake-up if there ever was one: the loop resumes)
                moved = False
                try:
                    moved = b.fill_certify(kind)
                except Exception as e:
                    err = e
                if not self._vote(moved, err):
                    break
                active = False
                continue
            try:
                active = b.fill_batch(kind) if busy else False
            except Exception as e:
                err = e
        b.fill_end(kind)

    def fill(self):
        """fill.fill_terrain over all bands (+ bluespot depths of the owned rows)."""
        self._fill(0, "filled")

    def short_and_diag(self):
        """fill.minimum_safe_short_and_diag on the GLOBAL raster (reference fill.py:235-250)."""
        mn, mx, _ = self.band.dem_minmax()
        amax = np.float32(max(self.comm.allgather(float(mx))))
        amin = np.float32(min(self.comm.allgather(float(mn))))
        maxval = np.float64(max(abs(amax), abs(amin)))
        nextval = np.nextafter(maxval, np.float64(float('inf')))
        short = (nextval - maxval) * 1024
        self.short, self.diag = float(short), float(short * (2 ** 0.5))
        return self.short, self.diag

    def noflat(self):
        """fill.fill_terrain_no_flats over all bands (needs fill())."""
        short, diag = self.short_and_diag()
        if hasattr(self.band, "geo_begin") and os.environ.get("MALSTROEM_BAND_NOFLAT", "") != "relaxation" and self._noflat_geodesic(short, diag):
            return
        self._fill(1, "noflat", short, diag)

    def _noflat_geodesic(self, short, diag):
        """The integer geodesic transform of csrc/noflat_geo.hip on bands: every band classifies and relaxes its tiles, then
        { neighbours trade the edge rows of the DISTANCES; a band whose halo row changed relaxes again } until nothing moves
        anywhere -- one exchange per crossing of a seam by a geodesic inside a flat, not one per batch of tile rounds.  Returns
        False (on every rank alike) when the path does not apply to this DEM or a band's final check fails: the caller then
        runs the float64 relaxation."""
        b = self.band
        self.exchanges["noflat"] = 0
        err, applicable, active = None, True, False
        try:
            applicable, active = b.geo_begin(short, diag)
        except Exception as e:
            err = e
        if self._vote(not applicable, err):          # some band holds a level without integer weights
            return False
        while True:
            busy = False
            try:
                ch_top, ch_bot = self._swap_edges("ngdist")
                self.exchanges["noflat"] += 1
                if ch_top:
                    b.geo_halo_changed(0)
                if ch_bot:
                    b.geo_halo_changed(1)
                busy = active or ch_top or ch_bot
            except Exception as e:
                err = e
            if not self._vote(busy, err):
                break
            try:
                if busy:
                    active = b.geo_batch()
            except Exception as e:
                err = e
        ok, partial = False, False
        try:
            ok, partial = b.geo_end()
        except Exception as e:
            err = e
        if self._vote(not ok, err):
            return False
        if not self._vote(partial):
            return True
        # some band holds flats the transform does not cover (a sea at elevation 0, a distance beyond the uint32 headroom): every
        # band relaxes in float64 from the surface it has -- exact, or an upper bound on those flats -- and checks the result
        self._fill(1, "noflat", short, diag, attach=True)
        try:
            ok = b.noflat_verify()
        except Exception as e:
            err = e
        return not self._vote(not ok, err)

    def flowdir(self):
        """D8 on the no-flats surface, edges outward; afterwards the flow-direction halo rows are valid too."""
        self.band.run_flowdir()
        self._swap_edges("flowdir")

    def accum(self):
        """flow.accumulated_flow over all bands in two local passes and one all-gather, however often the rivers cross the
        seams.  Pass 1 (``accum_boundary``) accumulates every band's OWN cells and finds, for each cell of a neighbour's edge
        row that flows into the band, the edge cell through which that flux leaves the band again; with it the values of all
        seam-crossing cells form a forest  x[e] = own[e] + sum(x[k] for the k that leave through e)  which every rank solves
        identically (``solve_band_accum``); pass 2 is the local accumulation with the solved halo rows as known sources."""
        b, W = self.band, self.W
        self.exchanges["accum"] = 0
        err, info = None, None
        try:
            exit_map = b.accum_boundary()
            info = dict(a0=(b.get_edge_row("accum", 0), b.get_edge_row("accum", 1)), exit=exit_map)
        except Exception as e:
            err = e
        self._vote(False, err)
        allinfo = self._cur_comm().allgather(info)
        self.exchanges["accum"] += 1
        try:
            top, bot = solve_band_accum(allinfo, W)[self.comm.rank]
            if self.has_up:
                b.set_halo_row("accum", 0, top)
            if self.has_down:
                b.set_halo_row("accum", 1, bot)
            b.run_accum()
        except Exception as e:
            err = e
        self._vote(False, err)

    def label(self):
        """label.connected_components over all bands with scipy's numbering (order of first raster pixel).

        Every band labels its local raster (owned + halo rows); the four boundary rows of every band are gathered,
        the cross-band equivalences are solved identically on every rank, and each band rewrites its labels through
        a LUT.  Returns the global number of labels."""
        import scipy.sparse
        import scipy.sparse.csgraph
        b, comm, W = self.band, self._cur_comm(), self.W
        lap = _Lap("label", comm.rank)
        nloc = b.ccl_local()
        lap.lap("ccl_local")
        rows = dict(nloc=nloc, first=b.get_edge_row("labels", 0), last=b.get_edge_row("labels", 1),
                    top=b.get_edge_row("labels", 2) if self.has_up else None,
                    bot=b.get_edge_row("labels", 3) if self.has_down else None)
        lap.lap("edge rows")
        allrows = comm.allgather(rows)
        lap.lap("allgather")
        R = comm.size
        key = lambda r, lab: (np.int64(r) << 32) | lab.astype(np.int64)
        # equivalences between band r and r+1: r.last == (r+1).top and r.bot == (r+1).first, cell by cell -- one edge per RUN of
        # equal (a, c) pairs along the row (the cells of a run repeat the same edge; everything below then works on a few
        # thousand edges instead of 2 W per seam: at W = 65536 and 8 bands this merge was 350 ms of host time on every rank)
        ea, eb = [], []
        for r in range(R - 1):
            up, dn = allrows[r], allrows[r + 1]
            for a, c in ((up["last"], dn["top"]), (up["bot"], dn["first"])):
                m = (a > 0) & (c > 0)
                if a.size > 1:
                    m[1:] &= (a[1:] != a[:-1]) | (c[1:] != c[:-1])
                ea.append(key(r, a[m]))
                eb.append(key(r + 1, c[m]))
        ea = np.concatenate(ea) if ea else np.zeros(0, np.int64)
        eb = np.concatenate(eb) if eb else np.zeros(0, np.int64)
        nodes, inv = np.unique(np.concatenate([ea, eb]), return_inverse=True)
        nn = nodes.size
        if nn:
            g = scipy.sparse.coo_matrix((np.ones(ea.size, np.int8), (inv[:ea.size], inv[ea.size:])), shape=(nn, nn))
            ncls, cls = scipy.sparse.csgraph.connected_components(g, directed=False)
        else:
            ncls, cls = 0, np.zeros(0, np.int64)
        node_rank = (nodes >> 32).astype(np.int64)
        node_lab = (nodes & 0xffffffff).astype(np.int64)
        # phantom = local component made of halo cells only (no owned cell): in a halo row but not in the adjacent owned row
        def run_values(row):
            """the distinct positive labels of a row, from its run starts (a row holds far fewer runs than cells)"""
            keep = row > 0
            if row.size > 1:
                keep[1:] &= row[1:] != row[:-1]
            return np.unique(row[keep])
        phantom = np.zeros(nn, bool)
        bounds = np.searchsorted(node_rank, np.arange(R + 1))     # `nodes` is sorted by (rank, label): rank r = one slice
        for r in range(R):
            rr = allrows[r]
            sl = slice(int(bounds[r]), int(bounds[r + 1]))
            for halo, edge in ((rr["top"], rr["first"]), (rr["bot"], rr["last"])):
                if halo is None:
                    continue
                ph = np.setdiff1d(run_values(halo), run_values(edge), assume_unique=True)
                if ph.size:
                    phantom[sl] |= np.isin(node_lab[sl], ph, assume_unique=True)
        # class owner = smallest rank with a real member; representative = smallest local label of the owner's members
        big = np.int64(1) << 62
        score = np.where(phantom, big, (node_rank << 32) | node_lab)
        rep_score = np.full(ncls, big, np.int64)
        np.minimum.at(rep_score, cls, score)
        is_rep = score == rep_score[cls]
        # ---- my numbering: all local labels except the DROPPED ones (phantoms and non-representative class members) keep
        # their order; kept local label l becomes offset + l - #(dropped labels < l).  Only the (short) dropped list is
        # ever materialised: the band's labels are rewritten on the device (relabel_sparse).
        mine = node_rank == comm.rank
        dropped = np.unique(node_lab[mine & ~is_rep]).astype(np.int64)
        n_own = int(nloc - dropped.size)
        offsets = np.concatenate([[0], np.cumsum(comm.allgather(n_own))])
        off = int(offsets[comm.rank])
        newlab = lambda l: off + l - np.searchsorted(dropped, l)      # for kept local labels l
        # owners publish the global label of the classes they own; members look it up
        my_reps = mine & is_rep
        published = comm.allgather((cls[my_reps], newlab(node_lab[my_reps])))
        class_label = np.zeros(ncls, np.int64)
        for ids, labs in published:
            class_label[ids] = labs
        members = mine & ~is_rep
        target = np.zeros(dropped.size, np.int64)                        # phantoms own no cell here: any value
        target[np.searchsorted(dropped, node_lab[members])] = class_label[cls[members]]
        # global labels with cells in more than one band (same array on every rank): their records need a merge
        real = ~phantom
        pairs = np.unique(cls[real].astype(np.int64) * R + node_rank[real]) if real.any() else np.zeros(0, np.int64)   # (class, rank)
        ranks_per_class = np.bincount(pairs // R, minlength=ncls) if ncls else np.zeros(0, np.int64)
        self.shared_labels = np.unique(class_label[np.flatnonzero(ranks_per_class > 1)]).astype(np.int64)
        self.nlabels = int(offsets[-1])
        self.label_range = (int(offsets[comm.rank]) + 1, int(offsets[comm.rank + 1]))
        lap.lap("merge (host)")
        b.relabel_sparse(nloc, off, dropped.astype(np.int32), target.astype(np.int32), self.nlabels)
        lap.lap("relabel_sparse")
        lap.report()
        return self.nlabels

    def watershed(self):
        """flow.watersheds_from_labels over all bands: local pointer jumping with pseudo labels on the halo rows, then the
        boundary system (2 rows per band) is solved identically on every rank and applied as a LUT."""
        b, comm, W = self.band, self._cur_comm(), self.W
        b.watershed_local()
        rows = comm.allgather((b.get_edge_row("watersheds", 0), b.get_edge_row("watersheds", 1)))
        R = comm.size
        # node(r, s, c) = (2r + s) * W + c for the first (s=0) / last (s=1) owned row of band r
        vals = np.concatenate([np.concatenate(p) for p in rows]).astype(np.int64)
        for r in range(R):
            for s_ in (0, 1):
                seg = vals[(2 * r + s_) * W:(2 * r + s_ + 1) * W]
                neg = seg < 0
                idx = -seg[neg] - 1
                up = idx < W          # pseudo label of band r's top halo = last owned row of band r-1
                tgt = np.where(up, (2 * (r - 1) + 1) * W + idx, (2 * (r + 1)) * W + (idx - W))
                seg[neg] = -(tgt + 1)
        vals = np.ascontiguousarray(vals)
        _lib.call("mhip_band_ws_resolve", _lib.i64(vals.size), _lib.ptr(vals))   # chains followed to their end; a flow cycle across bands: 0
        lut = np.zeros(2 * W, np.int64)
        if self.has_up:
            lut[:W] = vals[(2 * (comm.rank - 1) + 1) * W:(2 * (comm.rank - 1) + 2) * W]
        if self.has_down:
            lut[W:] = vals[(2 * (comm.rank + 1)) * W:(2 * (comm.rank + 1) + 1) * W]
        b.apply_neg_lut("watersheds", lut.astype(np.int32))

    # ---- per-label records (reference bluespots.py:159-206 on one raster).  Every rank returns the records of the labels
    # IT numbered (``label_range``, complete after the merge) plus the background record; a label or a watershed that
    # reaches into other bands is merged from their partial records (a handful of rows per exchange).
    def _cur_comm(self):
        """the communicator of the calling thread: the clone inside run_chain's labelling branch, else the main one"""
        return getattr(self._tls, "comm", None) or self.comm

    def _merged(self, which, merge, fetch_own=True):
        """own-range slice of record set `which`, with the labels that live in several bands (and the background) merged
        by `merge(list of per-rank record arrays) -> array`.  fetch_own=False: compute and merge only (the slice of this
        band's own labels stays on the device, like the records of the single-GPU pipeline until somebody asks for them)"""
        b, (lo, hi) = self.band, self.label_range
        lap = _Lap("records %d" % which, self.comm.rank)
        b.records_compute(which)
        lap.lap("compute")
        ids = np.concatenate([[0], self.shared_labels]).astype(np.int64)
        g = b.records_gather(which, ids)
        lap.lap("gather %d" % ids.size)
        parts = self._cur_comm().allgather(g)
        lap.lap("allgather")
        m = merge(parts)
        lap.lap("merge")
        lap.report()
        if not fetch_own:
            return {"first_label": lo, "records": None, "shared_labels": ids, "shared_records": m, "background": m[0].copy()}
        own = b.records_fetch(which, lo, hi - lo + 1)
        sel = (ids >= lo) & (ids <= hi)
        own[ids[sel] - lo] = m[sel]
        return {"first_label": lo, "records": own, "background": m[0].copy()}

    def stats(self, fetch_own=True):
        """label.label_stats(depths, labels): min / max / sum / count per bluespot."""
        def merge(parts):
            m = parts[0].copy()
            for p in parts[1:]:
                m["min"] = np.minimum(m["min"], p["min"])
                m["max"] = np.maximum(m["max"], p["max"])
                m["sum"] = m["sum"] + p["sum"]
                m["count"] = m["count"] + p["count"]
            return m
        return self._merged(0, merge, fetch_own)

    def watershed_counts(self, fetch_own=True):
        """label.label_count(watersheds): cells per watershed.  A watershed may reach into any band, so every rank publishes
        its non-zero counts of labels it did not number (a sparse handful) and adds what the others found of its own."""
        b, (lo, hi) = self.band, self.label_range
        b.records_compute(1)
        own = b.records_fetch(1, lo, hi - lo + 1) if fetch_own else None
        fid, fval = b.foreign_counts(lo, hi)
        bg = int(b.records_fetch(1, 0, 1)[0])
        total0 = 0
        for r, (ids, vals, c0) in enumerate(self._cur_comm().allgather((fid, fval, bg))):
            total0 += c0
            if r != self.comm.rank and own is not None:
                mine = (ids >= lo) & (ids <= hi)
                np.add.at(own, ids[mine] - lo, vals[mine])
        return {"first_label": lo, "records": own, "background": np.int64(total0)}

    def pourpoints(self, fetch_own=True):
        """label.label_max_index(accum, labels): value, row, col of the first raster cell with the largest accumulated
        flow per bluespot (rows are global)."""
        def merge(parts):
            m = parts[0].copy()
            for p in parts[1:]:
                # strict '>' with the first raster position on ties: bands are in raster order, so an earlier band wins ties
                better = (p["value"] > m["value"]) | ((m["row"] < 0) & (p["row"] >= 0))
                m[better] = p[better]
            return m
        return self._merged(2, merge, fetch_own)

    # ---- the whole chain with the stage DAG of mhip_ctx_run: labelling (+ stats) on a second host thread, a second
    # communicator and the band's side stream, next to no-flats fill -> D8 -> accumulation
    def run_chain(self, records=True, fetch_own=True, overlap=True, timings=None):
        """fill, no-flats fill, D8, accumulation, labels, watersheds (+ merged per-label records).  Returns the records dict
        (None entries when ``records`` is False).  ``timings``: optional dict that receives wall-clock ms per stage."""
        import time
        t = {} if timings is None else timings

        def timed(name, fn):
            t0 = time.perf_counter()
            r = fn()
            t[name] = t.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
            return r

        timed("fill", self.fill)
        out = {"stats": None, "counts": None, "pour": None}
        if overlap and not hasattr(self, "_comm_b"):
            self._comm_b = self.comm.clone()
        err = []

        def label_branch():
            try:
                if hasattr(self.band, "side_begin"):
                    self.band.side_begin()
                self._tls.comm = self._comm_b
                timed("label", self.label)
                if records:
                    out["stats"] = timed("label", lambda: self.stats(fetch_own))
            except Exception as e:      # re-raised on the main thread
                err.append(e)
            finally:
                self._tls.comm = None
                if hasattr(self.band, "side_end"):
                    try:
                        self.band.side_end()
                    except Exception as e:
                        err.append(e)

        def ws_branch():
            try:
                if hasattr(self.band, "side_begin"):
                    self.band.side_begin()
                self._tls.comm = self._comm_b
                timed("watershed", self.watershed)
                if records:
                    out["counts"] = timed("watershed", lambda: self.watershed_counts(fetch_own))
            except Exception as e:
                err.append(e)
            finally:
                self._tls.comm = None
                if hasattr(self.band, "side_end"):
                    try:
                        self.band.side_end()
                    except Exception as e:
                        err.append(e)

        if not overlap:
            timed("noflat", self.noflat)
            timed("flowdir", self.flowdir)
            timed("accum", self.accum)
            timed("label", self.label)
            if records:
                out["stats"] = timed("label", lambda: self.stats(fetch_own))
            timed("watershed", self.watershed)
            if records:
                out["counts"] = timed("watershed", lambda: self.watershed_counts(fetch_own))
                out["pour"] = timed("pourpoints", lambda: self.pourpoints(fetch_own))
            return out

        # the main thread makes many short library calls (a batch of rounds, a halo swap, a one-float all-reduce ...); with
        # CPython's default 5 ms switch interval every one of them could wait that long for the GIL while the side thread
        # runs NumPy code
        import sys
        old_switch = sys.getswitchinterval()
        sys.setswitchinterval(1e-4)
        def together(main_steps, side):
            """main_steps on this thread next to `side` on a second one; failures on either thread of ANY rank are voted
            on (host communicator) after the join, so that all ranks raise together instead of one rank leaving the others
            inside their next collective"""
            th = threading.Thread(target=side)
            th.start()
            try:
                for name, fn in main_steps:
                    timed(name, fn)
            except Exception as e:
                err.append(e)
            finally:
                th.join()
            if self.comm.allreduce_max(1.0 if err else 0.0) > 0.0:
                raise err[0] if err else RuntimeError("another band failed in this phase of the chain")

        try:
            if os.environ.get("MALSTROEM_BAND_LABEL_START", "noflat") == "fill":
                together([("noflat", self.noflat), ("flowdir", self.flowdir)], label_branch)
                together([("accum", self.accum)], ws_branch)
            else:
                # like the single-GPU DAG (csrc/api.hip, measured there): the no-flats fill has the GPU to itself -- its many small
                # launches queue behind the labelling's long workgroups otherwise -- then labelling + watersheds (the latter
                # need labels and flow directions) run next to D8 + accumulation
                timed("noflat", self.noflat)
                if self.comm.allreduce_max(0.0) > 0.0:      # (keeps the bands in step before the two-thread phase)
                    raise RuntimeError("another band failed in the no-flats fill")
                flow_ready = threading.Event()

                def flowdir_then_signal():
                    try:
                        self.flowdir()
                    finally:
                        flow_ready.set()

                def side():
                    label_branch()
                    flow_ready.wait()
                    if not err:
                        ws_branch()

                together([("flowdir", flowdir_then_signal), ("accum", self.accum)], side)
        finally:
            sys.setswitchinterval(old_switch)
        if records:
            out["pour"] = timed("pourpoints", lambda: self.pourpoints(fetch_own))
        return out

    def download(self, name):
        return self.band.download(name)
