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
import hashlib
import hmac
import json
import os
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


class _Rendezvous(object):
    """The meeting point of the n band threads of one process.  In the library where it can be loaded (``mhip_tg_*``: a thread blocks
    inside a ctypes call -- no GIL --, spins a few microseconds and then sleeps; 40-60 rendezvous per step of a band chain cost
    ~0.1 ms each as ``threading.Barrier`` / ``queue.Queue`` hand-overs with the device idle), as ``threading.Barrier`` elsewhere.
    A wait that outlasts ``timeout_s`` raises on every thread (a band that died must not leave the others waiting)."""

    def __init__(self, n, timeout_s=600.0):
        self.n, self.timeout_ms = int(n), int(timeout_s * 1000)
        self._tg = None
        if os.environ.get("MALSTROEM_THREAD_RENDEZVOUS", "") != "python":
            try:
                h = ctypes.c_void_p()
                _lib.call("mhip_tg_create", self.n, ctypes.byref(h))
                self._tg = h
            except Exception:
                self._tg = None
        self._barrier = threading.Barrier(self.n) if self._tg is None else None
        self._slots = [None] * self.n

    def __del__(self):
        try:
            if self._tg is not None:
                _lib.call("mhip_tg_destroy", self._tg)
        except Exception:
            pass

    def wait(self):
        if self._tg is not None:
            _lib.call("mhip_tg_barrier", self._tg, self.timeout_ms)
        else:
            self._barrier.wait(timeout=self.timeout_ms / 1e3)

    def gather(self, rank, obj):
        """every thread's object, in rank order"""
        self._slots[rank] = obj
        self.wait()
        out = list(self._slots)
        self.wait()
        return out

    def max(self, rank, value):
        if self._tg is None:
            return max(self.gather(rank, float(value)))
        out = ctypes.c_double(0.0)
        _lib.call("mhip_tg_allreduce_max", self._tg, int(rank), ctypes.c_double(float(value)), ctypes.byref(out), self.timeout_ms)
        return out.value

    def rows(self, rank, to_up, to_down):
        """neighbour exchange between the threads: (row of rank - 1 offered downwards or None, row of rank + 1 offered upwards or None)"""
        if self._tg is None:
            offers = self.gather(rank, (None if to_up is None else np.array(to_up, copy=True), None if to_down is None else np.array(to_down, copy=True)))
            return (offers[rank - 1][1] if rank > 0 else None), (offers[rank + 1][0] if rank < self.n - 1 else None)
        a = None if to_up is None else np.ascontiguousarray(to_up)
        b = None if to_down is None else np.ascontiguousarray(to_down)

        def describe(x):      # bytes, item size, dtype kind, dimensions, shape (up to four)
            if x.ndim > 4 or x.dtype.kind not in "fiub":
                raise ValueError("rows travel as plain numeric arrays of at most four dimensions")
            return (ctypes.c_int64 * 8)(x.nbytes, x.dtype.itemsize, ord(x.dtype.kind), x.ndim, *(list(x.shape) + [0] * (4 - x.ndim)))

        def buffer_for(meta):
            if meta[0] < 0:
                return None
            return np.empty(tuple(int(meta[4 + k]) for k in range(int(meta[3]))), dtype=np.dtype("%s%d" % (chr(int(meta[2])), int(meta[1]))))

        m_up, m_down = (ctypes.c_int64 * 8)(), (ctypes.c_int64 * 8)()
        _lib.call("mhip_tg_offer", self._tg, int(rank), _lib.ptr(a) if a is not None else None, describe(a) if a is not None else None,
                  _lib.ptr(b) if b is not None else None, describe(b) if b is not None else None, m_up, m_down, self.timeout_ms)
        from_up, from_down = buffer_for(m_up), buffer_for(m_down)
        _lib.call("mhip_tg_take", self._tg, int(rank), _lib.ptr(from_up) if from_up is not None else None,
                  _lib.ptr(from_down) if from_down is not None else None, self.timeout_ms)
        return from_up, from_down


class ThreadComm(Comm):
    """In-process transport: ``ThreadComm.world(n)`` returns n endpoints to be driven by n threads."""

    class _World(object):
        def __init__(self, n):
            self.n = n
            self.meet = _Rendezvous(n)

    @classmethod
    def world(cls, n):
        w = cls._World(n)
        return [cls(w, r) for r in range(n)]

    def __init__(self, world, rank):
        self._w = world
        self.rank = rank
        self.size = world.n

    def exchange_rows(self, to_up, to_down):
        return self._w.meet.rows(self.rank, to_up if self.rank > 0 else None, to_down if self.rank < self.size - 1 else None)

    def allgather(self, obj):
        return self._w.meet.gather(self.rank, obj)

    def allreduce_max(self, value):
        return self._w.meet.max(self.rank, value)

    def clone(self):
        w = self.allgather(ThreadComm._World(self.size) if self.rank == 0 else None)[0]   # rank 0's object, shared in-process
        return ThreadComm(w, self.rank)


# ---- wire format of SocketComm: data only.  Nothing that arrives on the socket is ever executed or imported (no pickle): a
# message is a tree of None / bool / int / float / str / bytes / list / tuple / dict / numpy arrays and scalars (structured record
# arrays included), written as one tag byte per node, little-endian lengths and the raw array bytes.
def _pack(obj, out):
    if obj is None:
        out.append(b"N")
    elif isinstance(obj, (bool, np.bool_)):
        out.append(b"T" if obj else b"F")
    elif isinstance(obj, (int, np.integer)) and -(1 << 63) <= int(obj) < (1 << 63):
        out.append(b"i" + struct.pack("<q", int(obj)))
    elif isinstance(obj, (float, np.floating)):
        out.append(b"d" + struct.pack("<d", float(obj)))
    elif isinstance(obj, str):
        b = obj.encode("utf-8")
        out.append(b"s" + struct.pack("<q", len(b)) + b)
    elif isinstance(obj, (bytes, bytearray)):
        out.append(b"b" + struct.pack("<q", len(obj)) + bytes(obj))
    elif isinstance(obj, (list, tuple)):
        out.append((b"l" if isinstance(obj, list) else b"t") + struct.pack("<q", len(obj)))
        for o in obj:
            _pack(o, out)
    elif isinstance(obj, dict):
        out.append(b"m" + struct.pack("<q", len(obj)))
        for k, v in obj.items():
            _pack(k, out)
            _pack(v, out)
    elif isinstance(obj, np.ndarray):
        if obj.dtype.hasobject:
            raise TypeError("SocketComm: object arrays do not travel")
        a = np.ascontiguousarray(obj)
        head = json.dumps([a.dtype.descr if a.dtype.names else a.dtype.str, list(a.shape)]).encode("ascii")
        out.append(b"a" + struct.pack("<qq", len(head), a.nbytes) + head)
        out.append(a.tobytes())
    else:
        raise TypeError("SocketComm cannot send a %s" % type(obj).__name__)


def _unpack(buf, pos=0):
    tag = buf[pos:pos + 1]
    pos += 1
    if tag == b"N":
        return None, pos
    if tag in (b"T", b"F"):
        return tag == b"T", pos
    if tag == b"i":
        return struct.unpack_from("<q", buf, pos)[0], pos + 8
    if tag == b"d":
        return struct.unpack_from("<d", buf, pos)[0], pos + 8
    if tag in (b"s", b"b"):
        n = struct.unpack_from("<q", buf, pos)[0]
        raw = bytes(buf[pos + 8:pos + 8 + n])
        if n < 0 or len(raw) != n:
            raise ValueError("SocketComm: truncated message")
        return (raw.decode("utf-8") if tag == b"s" else raw), pos + 8 + n
    if tag in (b"l", b"t"):
        n = struct.unpack_from("<q", buf, pos)[0]
        pos += 8
        items = []
        for _ in range(n):
            o, pos = _unpack(buf, pos)
            items.append(o)
        return (items if tag == b"l" else tuple(items)), pos
    if tag == b"m":
        n = struct.unpack_from("<q", buf, pos)[0]
        pos += 8
        d = {}
        for _ in range(n):
            k, pos = _unpack(buf, pos)
            v, pos = _unpack(buf, pos)
            d[k] = v
        return d, pos
    if tag == b"a":
        nh, nb = struct.unpack_from("<qq", buf, pos)
        pos += 16
        descr, shape = json.loads(bytes(buf[pos:pos + nh]).decode("ascii"))
        pos += nh
        dtype = np.dtype([tuple(f) for f in descr]) if isinstance(descr, list) else np.dtype(descr)
        if dtype.hasobject or nb != dtype.itemsize * int(np.prod(shape, dtype=np.int64)) or pos + nb > len(buf):
            raise ValueError("SocketComm: malformed array")
        a = np.frombuffer(buf, dtype=dtype, count=nb // dtype.itemsize if dtype.itemsize else 0, offset=pos).reshape(shape).copy()
        return a, pos + nb
    raise ValueError("SocketComm: unknown tag %r" % tag)


def wire_dumps(obj):
    out = []
    _pack(obj, out)
    return b"".join(out)


def wire_loads(buf):
    obj, pos = _unpack(memoryview(buf).tobytes() if not isinstance(buf, bytes) else buf, 0)
    if pos != len(buf):
        raise ValueError("SocketComm: trailing bytes")
    return obj


def _is_loopback(addr):
    a = str(addr).strip().lower()
    return a in ("localhost", "::1", "") or a.startswith("127.")


class SocketComm(Comm):
    """Control-plane transport over plain TCP sockets (stdlib only): rank 0 listens on (addr, port), every other rank
    connects once; a collective is "send my object to rank 0, receive the list of everybody's" in the data-only wire format above
    (never pickle: whoever reaches the port cannot make a rank execute anything).  The handshake carries the rank; rank 0 refuses
    ranks outside 1 .. size-1 and duplicates.  With a shared secret (``secret=`` or MALSTROEM_COMM_SECRET in the launcher's
    environment) every connection must answer an HMAC-SHA256 challenge first.  Messages are a
    few rows at most (the boundary systems of the label / watershed protocols, the 128-byte ncclUniqueId), so the star
    topology is not a bottleneck; the rows of the fills travel over RCCL, not through here.

    ``SocketComm.from_env()`` reads RANK / WORLD_SIZE / MASTER_ADDR / MALSTROEM_COMM_PORT (default MASTER_PORT + 17)."""

    one_rank_per_device = True   # ranks are processes, by convention one per GPU: BandPipeline may put the rows on RCCL

    def __init__(self, rank, size, addr="127.0.0.1", port=29517, timeout_s=300.0, _socks=None, secret=None):
        self.rank, self.size, self._addr, self._port, self._timeout = int(rank), int(size), addr, int(port), float(timeout_s)
        if secret is None:
            secret = os.environ.get("MALSTROEM_COMM_SECRET")
        self._secret = secret.encode("utf-8") if isinstance(secret, str) else secret
        # Without a secret the HMAC key is empty, i.e. anybody who reaches the port can join as a rank (and feed halo rows or the
        # ncclUniqueId): fine on the loopback interface, refused on any other address
        if int(size) > 1 and not self._secret and not _is_loopback(addr):
            raise ValueError("SocketComm on %r needs a shared secret (MALSTROEM_COMM_SECRET, or secret=...): without one any host that "
                             "reaches the port could join as a rank" % (addr,))
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
                    try:
                        conn.settimeout(self._timeout)
                        conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        nonce = os.urandom(32)
                        conn.sendall(nonce)
                        r = struct.unpack("<i", self._recv_exact(conn, 4))[0]
                        mac = self._recv_exact(conn, 32)
                        want = hmac.new(self._secret or b"", nonce + struct.pack("<i", r), hashlib.sha256).digest()
                        if not (1 <= r < self.size) or r in self._peers or not hmac.compare_digest(mac, want):
                            raise ConnectionError("SocketComm: connection refused (rank %d, duplicate or bad credentials)" % r)
                    except (OSError, struct.error, ConnectionError):
                        conn.close()        # a stray or hostile connection: drop it and keep listening for the real ranks
                        continue
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
            nonce = self._recv_exact(self._hub, 32)
            me = struct.pack("<i", self.rank)
            self._hub.sendall(me + hmac.new(self._secret or b"", nonce + me, hashlib.sha256).digest())

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

    # the largest message a peer may announce (the seam systems gather a few MB, `gather_rows` a window of rows): a length word
    # beyond this is a broken or hostile peer, not something to allocate
    MAX_MESSAGE = int(os.environ.get("MALSTROEM_COMM_MAX_MESSAGE", 8 << 30))

    @classmethod
    def _recv_msg(cls, sock):
        n = struct.unpack("<q", cls._recv_exact(sock, 8))[0]
        if n < 0 or n > cls.MAX_MESSAGE:
            raise ConnectionError("SocketComm: a peer announced a message of %d bytes (limit %d, MALSTROEM_COMM_MAX_MESSAGE)" % (n, cls.MAX_MESSAGE))
        return cls._recv_exact(sock, n)

    def allgather(self, obj):
        if self.size == 1:
            return [obj]
        with self._lock:
            if self.rank == 0:
                parts = [None] * self.size
                parts[0] = obj
                for r, conn in self._peers.items():
                    parts[r] = wire_loads(self._recv_msg(conn))
                blob = wire_dumps(parts)
                for conn in self._peers.values():
                    self._send_msg(conn, blob)
                return parts
            self._send_msg(self._hub, wire_dumps(obj))
            return wire_loads(self._recv_msg(self._hub))

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
        return SocketComm(self.rank, self.size, self._addr, self._port + self._nclones, self._timeout, secret=self._secret)

    def close(self):
        for c in list(self._peers.values()) + ([self._hub] if self._hub else []):
            try:
                c.close()
            except OSError:
                pass
        self._peers, self._hub = {}, None


class ShmComm(Comm):
    """The control plane between ranks that are PROCESSES OF ONE HOST, in a POSIX shared-memory segment: a collective is "write my
    payload into my slot, barrier, read the others', barrier" -- two barriers of a few microseconds (``mhip_shm_barrier``: the
    hardware's atomics on two words of the segment) instead of a TCP round trip through gloo or the socket hub (0.15-0.5 ms each,
    50-60 per step of a band chain).  Payloads travel in the data-only wire format (``wire_dumps``: nothing that arrives is ever
    executed), in pieces of the slot size where they are larger.  ``ShmComm.over(comm)`` sets one up through an existing
    communicator (a collective call: rank 0 creates the segment, the others attach; every rank must be on rank 0's host --
    checked by a file every rank looks for in the segment's directory -- or every rank gets ``comm`` back)."""

    one_rank_per_device = True
    HEADER = 4096

    def __init__(self, rank, size, shm, slot_bytes, owner, parent, timeout_s=600.0):
        self.rank, self.size = int(rank), int(size)
        self._shm, self._slot, self._owner, self._parent = shm, int(slot_bytes), bool(owner), parent
        self._timeout_ms = int(timeout_s * 1000)
        buf = shm.buf
        self._base = ctypes.addressof(ctypes.c_char.from_buffer(buf))
        self._values = np.frombuffer(buf, dtype=np.float64, count=self.size, offset=256)
        self._lens = np.frombuffer(buf, dtype=np.int64, count=self.size, offset=256 + 8 * self.size + (-8 * self.size) % 64)
        self._slots = [np.frombuffer(buf, dtype=np.uint8, count=self._slot, offset=self.HEADER + r * self._slot) for r in range(self.size)]

    @classmethod
    def over(cls, comm, slot_bytes=8 << 20, timeout_s=600.0):
        """collective on ``comm``: a ShmComm of the same ranks, or ``comm`` itself where the ranks do not share a host / the segment
        cannot be made (decided together)"""
        from multiprocessing import shared_memory
        if comm.size < 2:
            return comm
        total = cls.HEADER + comm.size * int(slot_bytes)
        shm, name, err = None, None, None
        if comm.rank == 0:
            try:
                shm = shared_memory.SharedMemory(create=True, size=total)
                shm.buf[:cls.HEADER] = bytes(cls.HEADER)
                name = shm.name
            except Exception as e:      # (no /dev/shm, no room ...)
                err = e
        name = comm.allgather(name)[0]
        if name is not None and comm.rank != 0:
            try:
                shm = shared_memory.SharedMemory(name=name)      # (only a process of rank 0's host finds it)
                if shm.size < total:
                    raise RuntimeError("segment too small")
            except Exception as e:
                err, shm = e, None
        ok = comm.allreduce_max(0.0 if (shm is not None and err is None) else 1.0) == 0.0
        if not ok:
            if shm is not None:
                shm.close()
                if comm.rank == 0:
                    shm.unlink()
            return comm
        if comm.rank != 0:
            # (the resource tracker of a process that merely attached would unlink the segment when that process ends -- Python < 3.13)
            try:
                from multiprocessing import resource_tracker
                resource_tracker.unregister(shm._name, "shared_memory")
            except Exception:
                pass
        return cls(comm.rank, comm.size, shm, slot_bytes, comm.rank == 0, comm, timeout_s)

    def _barrier(self):
        _lib.call("mhip_shm_barrier", ctypes.c_void_p(self._base), self.size, self._timeout_ms)

    def allreduce_max(self, value):
        self._values[self.rank] = float(value)
        self._barrier()
        m = float(self._values.max())
        self._barrier()
        return m

    def _trade(self, payload, wanted):
        """my bytes into my slot, the bytes of the ranks in ``wanted`` out of theirs -> {rank: bytes}; in pieces where a payload is
        larger than a slot"""
        n = len(payload)
        self._lens[self.rank] = n
        self._barrier()
        lens = self._lens.copy()
        out = {r: bytearray() for r in wanted}
        mine = np.frombuffer(payload, dtype=np.uint8) if n else None
        for off in range(0, max(int(lens.max()), 1), self._slot):
            k = max(0, min(self._slot, n - off))
            if k:
                self._slots[self.rank][:k] = mine[off:off + k]
            self._barrier()
            for r in wanted:
                kr = max(0, min(self._slot, int(lens[r]) - off))
                if kr:
                    out[r] += self._slots[r][:kr].tobytes()
            self._barrier()
        return {r: bytes(b) for r, b in out.items()}

    def allgather(self, obj):
        got = self._trade(wire_dumps(obj), range(self.size))
        return [wire_loads(got[r]) for r in range(self.size)]

    def exchange_rows(self, to_up, to_down):
        up, down = self.rank - 1, self.rank + 1
        got = self._trade(wire_dumps((None if up < 0 else to_up, None if down >= self.size else to_down)),
                          [r for r in (up, down) if 0 <= r < self.size])
        from_up = wire_loads(got[up])[1] if up >= 0 else None
        from_down = wire_loads(got[down])[0] if down < self.size else None
        return from_up, from_down

    def clone(self):
        return ShmComm.over(self, self._slot, self._timeout_ms / 1e3)

    def close(self):
        try:
            self._values = self._lens = self._slots = None
            self._shm.close()
            if self._owner:
                self._shm.unlink()
        except Exception:
            pass


class HybridComm(Comm):
    """``k`` bands per process (threads) times ``P`` processes = ``P * k`` virtual ranks, band order = (process, thread).

    A GPU addresses at most 2**31 - 2 cells per band context (int32 cell indices keep the union-find / pointer-jumping
    rasters at 4 bytes per cell), so a raster beyond that runs as several bands on ONE GPU as well: the 1- and 2-GPU
    points of a 65536 x 65536 strong-scaling curve are 4 and 2 bands per process.  Several bands share a GPU here, so
    there is no RCCL communicator (one rank per device); objects travel through the process-level ``Comm`` (all-gathered), rows
    between neighbours only (in memory inside a process, through the process-level neighbour exchange at its two ends).

    ``HybridComm.world(proc_comm, k)`` -> the ``k`` endpoints of this process, one per band thread."""

    class _World(object):
        def __init__(self, proc, k):
            self.proc, self.k = proc, k
            self.meet = _Rendezvous(k, timeout_s=1800.0)
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
        w.meet.wait()
        if self._t == 0:
            w.result = [o for part in w.proc.allgather(list(w.slots)) for o in part]
        w.meet.wait()
        out = w.result
        w.meet.wait()
        return out

    def allreduce_max(self, value):
        """the threads' maximum, then ONE number through the process-level communicator (not an all-gather of objects)"""
        w = self._w
        w.slots[self._t] = float(value)
        w.meet.wait()
        if self._t == 0:
            w.result = w.proc.allreduce_max(max(w.slots)) if w.proc.size > 1 else max(w.slots)
        w.meet.wait()
        out = w.result
        w.meet.wait()
        return out

    def exchange_rows(self, to_up, to_down):
        """Neighbours only: the bands of one process trade their rows in memory, and only the process's FIRST band's row up and its
        LAST band's row down travel through the process-level communicator (its own neighbour exchange).  (Until round 4 every
        exchange all-gathered every band's rows as objects: 2 processes x 2 bands of 32768 columns, 132 instead of 116 ms a step.)"""
        w, t, k = self._w, self._t, self._w.k
        w.slots[t] = (None if to_up is None else np.array(to_up, copy=True), None if to_down is None else np.array(to_down, copy=True))
        w.meet.wait()
        if t == 0:
            if w.proc.size > 1:
                w.result = w.proc.exchange_rows(w.slots[0][0] if w.proc.rank > 0 else None,
                                                w.slots[k - 1][1] if w.proc.rank < w.proc.size - 1 else None)
            else:
                w.result = (None, None)
        w.meet.wait()
        from_up = w.slots[t - 1][1] if t > 0 else w.result[0]
        from_down = w.slots[t + 1][0] if t < k - 1 else w.result[1]
        w.meet.wait()        # (everybody has read: the slots may be written again)
        return from_up, from_down

    def clone(self):
        w = self._w
        w.meet.wait()
        if self._t == 0:
            w.result = HybridComm._World(w.proc.clone(), w.k)
            w.result.owns_proc = True
        w.meet.wait()
        nw = w.result
        w.meet.wait()
        return HybridComm(nw, self._t)

    def close(self):
        """(a clone's process-level communicator is the clone's own: its first endpoint closes it)"""
        w = self._w
        if self._t == 0 and getattr(w, "owns_proc", False) and hasattr(w.proc, "close"):
            w.owns_proc = False
            w.proc.close()


# ---- compute backend: one band context on one GPU -----------------------------------------------------------------

class HipBand(object):
    """ctypes face of a band ``mhip_ctx`` (include/malstroem_hip.h, row-band protocol)."""

    def __init__(self, H_global, W, row0, nrows, device=0, rank=0, size=1, unique_id=None):
        """``unique_id``: the 128 bytes of rank 0's ``HipBand.new_unique_id()``; with it the context joins the RCCL
        communicator of all ``size`` bands (a collective call) and moves its halo rows itself."""
        self.W, self.nrows, self.row0, self.H_global = int(W), int(nrows), int(row0), int(H_global)
        self.has_side_comm = False
        self._ctx = ctypes.c_void_p()
        uid = None
        if unique_id is not None:
            if len(unique_id) != 128:
                raise ValueError("ncclUniqueId must be 128 bytes")
            uid = ctypes.create_string_buffer(bytes(unique_id), 128)
        _lib.call("mhip_ctx_create_band", ctypes.byref(self._ctx), _lib.i64(H_global), _lib.i64(W), _lib.i64(row0),
                  _lib.i64(nrows), int(device), int(rank), int(size), uid)
        self.has_comm = int(_lib.load().mhip_ctx_has_comm(self._ctx)) >= 1

    @staticmethod
    def comm_available():
        """can this process load librccl with every symbol the band transport needs?  (no device call, no collective)"""
        return bool(_lib.load().mhip_comm_available())

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

    def exchange_edge_rows(self, name):
        """RCCL neighbour exchange of raster ``name``'s edge rows into HOST arrays, halo rows untouched -> (from_up, from_down);
        on the thread between side_begin / side_end it runs over the side communicator (add_side_comm)."""
        dt = RASTER_DTYPE[RASTERS[name]]
        up = np.empty(self.W, dtype=dt) if self.row0 > 0 else None
        dn = np.empty(self.W, dtype=dt) if self.row0 + self.nrows < self.H_global else None
        _lib.call("mhip_ctx_exchange_edge_rows", self._ctx, RASTERS[name], None if up is None else _lib.ptr(up), None if dn is None else _lib.ptr(dn))
        return up, dn

    def add_side_comm(self, unique_id):
        """a second RCCL communicator for the labelling thread (collective call)"""
        _lib.call("mhip_ctx_comm_add_side", self._ctx, ctypes.create_string_buffer(bytes(unique_id), 128))
        self.has_side_comm = int(_lib.load().mhip_ctx_has_comm(self._ctx)) == 2

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

    def download_rows(self, name, row0, nrows):
        """rows [row0, row0 + nrows) of the band's OWNED rows (a window on the host, whatever the band's height)"""
        which = RASTERS[name]
        out = np.empty((int(nrows), self.W), dtype=RASTER_DTYPE[which])
        if nrows:
            _lib.call("mhip_ctx_download_rows", self._ctx, which, _lib.i64(row0), _lib.i64(nrows), _lib.ptr(out))
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

    def get_edge_rows(self, name, first=True, last=True):
        """(first owned row or None, last owned row or None): one library call, one synchronisation"""
        which = RASTERS[name]
        a = np.empty(self.W, dtype=RASTER_DTYPE[which]) if first else None
        b = np.empty(self.W, dtype=RASTER_DTYPE[which]) if last else None
        _lib.call("mhip_ctx_get_edge_rows", self._ctx, which, _lib.ptr(a) if first else None, _lib.ptr(b) if last else None)
        return a, b

    def set_halo_rows(self, name, top, bottom):
        """the neighbours' rows into the halo rows (None: no neighbour); (top changed, bottom changed): one call, one synchronisation"""
        which = RASTERS[name]
        t = None if top is None else np.ascontiguousarray(top, dtype=RASTER_DTYPE[which])
        b = None if bottom is None else np.ascontiguousarray(bottom, dtype=RASTER_DTYPE[which])
        ch = (ctypes.c_int32 * 2)(0, 0)
        _lib.call("mhip_ctx_set_halo_rows", self._ctx, which, _lib.ptr(t) if t is not None else None, _lib.ptr(b) if b is not None else None, ch)
        return bool(ch[0]), bool(ch[1])

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

    def ccl_begin(self):
        """ccl_local without its emit pass: the number of band-local labels; of the labels raster only the edge rows exist until
        ``ccl_finish`` (csrc/api.hip: mhip_ctx_band_ccl_begin)"""
        n = ctypes.c_int64(0)
        _lib.call("mhip_ctx_band_ccl_begin", self._ctx, ctypes.byref(n))
        return int(n.value)

    def ccl_finish(self, offset, dropped, target, nlabels_global, with_stats):
        """the global label of every cell in one pass (the map of ``relabel_sparse``); with_stats: the label statistics of the owned
        rows on the same pass -- ``records_compute(0)`` need not run then"""
        d = np.ascontiguousarray(dropped, dtype=np.int32)
        t = np.ascontiguousarray(target, dtype=np.int32)
        _lib.call("mhip_ctx_band_ccl_finish", self._ctx, _lib.i64(offset), _lib.ptr(d), _lib.ptr(t), _lib.i64(d.size), _lib.i64(nlabels_global),
                  1 if with_stats else 0)

    def relabel(self, lut, nlabels_global):
        lut = np.ascontiguousarray(lut, dtype=np.int32)
        _lib.call("mhip_ctx_band_relabel", self._ctx, _lib.ptr(lut), _lib.i64(lut.size - 1), _lib.i64(nlabels_global))

    def relabel_sparse(self, nlocal, offset, dropped, target, nlabels_global):
        """local label l -> offset + l - #(dropped < l); dropped[k] (sorted) -> target[k]"""
        d = np.ascontiguousarray(dropped, dtype=np.int32)
        t = np.ascontiguousarray(target, dtype=np.int32)
        _lib.call("mhip_ctx_band_relabel_sparse", self._ctx, _lib.i64(nlocal), _lib.i64(offset), _lib.ptr(d), _lib.ptr(t), _lib.i64(d.size),
                  _lib.i64(nlabels_global))

    def relabel_range(self, lo, hi, lut, foreign_ids, foreign_new, nlabels_new):
        """labels lo..hi -> lut[l - lo]; labels numbered by other bands (sorted ``foreign_ids``) -> ``foreign_new``"""
        lut = np.ascontiguousarray(lut, dtype=np.int32)
        fid = np.ascontiguousarray(foreign_ids, dtype=np.int32)
        fnew = np.ascontiguousarray(foreign_new, dtype=np.int32)
        _lib.call("mhip_ctx_band_relabel_range", self._ctx, _lib.i64(lo), _lib.i64(hi), _lib.ptr(lut), _lib.ptr(fid), _lib.ptr(fnew), _lib.i64(fid.size),
                  _lib.i64(nlabels_new))

    def trace(self, cells, src, background_label, geometry):
        """one leg of the stream walk for walkers standing on this band's owned rows (global cells) ->
        (label, status, src, exit_cells, geoms): status 0 ended, 1 found ``label``, 2 handed over at ``exit_cells``"""
        cells = np.ascontiguousarray(np.asarray(cells, dtype=np.int64).reshape(-1, 2))
        n = cells.shape[0]
        src = np.ascontiguousarray(src, dtype=np.int32)
        use_bg = background_label is not None
        lab, status, src_out = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
        exits, lens = np.zeros((n, 2), np.int64), np.zeros(n, np.int64)

        def call(offsets, out_cells):
            _lib.call("mhip_ctx_band_trace", self._ctx, _lib.ptr(cells), _lib.ptr(src), _lib.i64(n), int(use_bg),
                      ctypes.c_int32(int(background_label) if use_bg else 0), _lib.ptr(lab), _lib.ptr(status), _lib.ptr(src_out), _lib.ptr(exits),
                      _lib.ptr(lens), None if offsets is None else _lib.ptr(offsets), None if out_cells is None else _lib.ptr(out_cells))
        call(None, None)
        geoms = [None] * n
        if geometry and n:
            offsets = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(lens, out=offsets[1:])
            flat = np.zeros(max(int(offsets[-1]), 1), dtype=np.int64)
            call(offsets, flat)
            geoms = [flat[int(offsets[i]):int(offsets[i + 1])] for i in range(n)]
        return lab, status, src_out, exits, geoms

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
    _REC_DTYPE = (STAT_DTYPE, np.dtype(np.int64), INDEX_DTYPE, INDEX_DTYPE)      # which = 0 stats, 1 watershed counts, 2 / 3 pour points (arg-max accum / arg-min no-flats)

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


_DEV = os.environ.get("MHIP_DEVELOPER") == "1"      # development knobs are only read under MHIP_DEVELOPER=1 (like the library's dev_env)
_ACCUM_AT_ONCE = _DEV and os.environ.get("MALSTROEM_BAND_ACCUM") == "at_once"      # A/B: the accumulation does not wait for the labelling's tile pass
_LABEL_THREE_PASSES = _DEV and os.environ.get("MALSTROEM_BAND_LABEL") == "passes"    # A/B: ccl_local + relabel_sparse + records_compute(0)
_PROF = _DEV and os.environ.get("MALSTROEM_BAND_PROFILE") is not None    # wall clock of the steps inside label() / _merged()


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
            tot, cnt = {}, {}
            for k, v in self.acc:       # (a name that recurs -- the laps of a loop -- is reported as its sum and count)
                tot[k] = tot.get(k, 0.0) + v
                cnt[k] = cnt.get(k, 0) + 1
            print("[band profile] %s: %s" % (self.what, ", ".join("%s %.1f%s" % (k, tot[k], " (%d x)" % cnt[k] if cnt[k] > 1 else "") for k in tot)),
                  file=sys.stderr, flush=True)


# one pass-through pair of the band accumulation (csrc/bandsolve.hip: BandAccumPair)
ACCUM_PAIR_DTYPE = np.dtype([("child", "<i4"), ("parent", "<i4"), ("own_child", "<f8"), ("own_parent", "<f8")])


class BandPipeline(object):
    """One rank's share of a row-banded DEM.  Every rank calls the same methods in the same order (SPMD)."""

    def __init__(self, comm, shape, device=0, backend_factory=None, rccl=None, align=True, rccl_side=False):
        """``align``: seams on the tile grid of the fill kernels (see ``band_rows``).  ``rccl``: True = the bands join an RCCL communicator inside the library and move their halo rows GPU -> GPU
        (one rank per GPU required), False = rows travel through ``comm`` as host buffers, None = RCCL whenever there is
        more than one rank and the backend is the HIP one.  Creation is voted on: if any rank fails, every rank raises.
        ``rccl_side``: the labelling thread of ``run_chain(overlap=True)`` gets an RCCL communicator of its own.  Off by default:
        two communicators of one device driven from two host threads are only safe when every rank issues their operations in
        the same order, which host-thread timing does not promise -- until that has run on two or more GPUs the labelling thread
        trades its edge rows through the host communicator (a clone of ``comm``), whose order is its own."""
        self.comm = comm
        self.H, self.W = int(shape[0]), int(shape[1])
        self.row0, self.nrows = band_rows(self.H, comm.size, comm.rank, align)
        factory = backend_factory or HipBand
        if rccl is None:   # RCCL wants one rank per device: only transports between processes say so (SocketComm; a launcher's own)
            rccl = comm.size > 1 and factory is HipBand and bool(getattr(comm, "one_rank_per_device", False))
        self.rccl_error = None
        for attempt in (0, 1):
            self.band, err = None, None
            uid = uid2 = None
            try:
                if rccl and comm.size > 1:
                    # every rank first says whether it can load RCCL at all (a cheap probe, no collective inside): a rank that
                    # cannot would return from the band creation at once while the others wait for it in ncclCommInitRank
                    probe = getattr(factory, "comm_available", None)      # (stand-in backends without the probe: always available)
                    if comm.allreduce_max(0.0 if (probe is None or probe()) else 1.0) > 0.0:
                        raise RuntimeError("RCCL cannot be loaded on at least one rank")
                    try:
                        mine = getattr(factory, "new_unique_id", HipBand.new_unique_id)() if comm.rank == 0 else None
                    except Exception as e:
                        mine, err = None, e
                    mine2 = None
                    if mine is not None and rccl_side:      # a second id: the labelling thread gets a communicator of its own
                        try:
                            mine2 = getattr(factory, "new_unique_id", HipBand.new_unique_id)()
                        except Exception:
                            mine2 = None
                    uid, uid2 = comm.allgather((mine, mine2))[0]          # every rank takes part, whatever happened on rank 0
                    if uid is None:
                        raise err or RuntimeError("rank 0 could not create an ncclUniqueId")
                kw = dict(unique_id=uid) if uid is not None else {}
                self.band = factory(self.H, self.W, self.row0, self.nrows, device=device, rank=comm.rank, size=comm.size, **kw)
                if uid is not None:                          # one collective before anything depends on the communicator
                    self.band.allreduce_max(float(comm.rank))
            except Exception as e:
                err = err or e
            if err is None and uid is not None and uid2 is not None and hasattr(self.band, "add_side_comm"):
                # ncclCommInitRank is a collective: every rank says it got this far BEFORE any rank enters it (a rank that failed
                # above would leave the others waiting inside); on a "no" the setup vote below takes every rank out together
                if comm.allreduce_max(0.0) == 0.0:
                    try:
                        self.band.add_side_comm(uid2)
                    except Exception as e:
                        err = e
            elif uid is not None and uid2 is not None:
                comm.allreduce_max(1.0)                      # (this rank failed: the others must not enter the collective)
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
        self.rccl_side = self.rccl and bool(getattr(self.band, "has_side_comm", False))
        self.host_ms = {}      # thread CPU time of the host-serial sections (boundary systems), by stage: see _host()
        self.has_up = comm.rank > 0
        self.has_down = comm.rank < comm.size - 1
        self.exchanges = {"fill": 0, "noflat": 0}
        self.short = self.diag = None
        self._tls = threading.local()

    def close(self):
        self.band.close()
        cb = getattr(self, "_comm_b", None)      # the labelling thread's communicator is the pipeline's own (run_chain cloned it)
        if cb is not None and hasattr(cb, "close"):
            try:
                cb.close()
            except Exception:
                pass
            self._comm_b = None

    def engines(self):
        """Which engine the last fill / no-flats fill of this band ran (``ctx_get_int``: fill 1 = tiled priority-flood, 0 = iterative
        schedule, 4 = flood + iterative repair; no-flats 2 = integer geodesic transform, 3 = + float64 relaxation of irregular flats,
        0 = float64 relaxation; accumulation 1 = the second pass as a delta over the boundary pass's perimeter graph, 0 = a full second
        pass): a silent fall-back is 2-3x slower and ``bench.py`` refuses to report it.  ``None``: a stand-in backend that does not say."""
        out = {}
        for name, key in (("fill", "fill_algorithm"), ("noflat", "noflat_algorithm"), ("accum", "accum_algorithm")):
            try:
                out[name] = int(self.band.get_int(key))
            except Exception:
                out[name] = None
        return out

    # ---- helpers
    def _allreduce_max(self, value):
        """the one-number reduction of the iteration loops: RCCL when the band owns a communicator -- unless the control plane is a
        shared-memory segment of the node (two barriers of a few microseconds: cheaper than a collective launch + a host
        synchronisation, and the value is the host's anyway)"""
        if self.rccl and not isinstance(self.comm, ShmComm):
            return self.band.allreduce_max(value)
        return self.comm.allreduce_max(value)

    def _vote(self, busy, err=None):
        """all-reduce of "still busy" that also carries failures: a rank that caught an exception votes 2 and every
        rank leaves the loop with an error instead of waiting for the failed one in the next collective"""
        m = self._allreduce_max(2.0 if err is not None else (1.0 if busy else 0.0))
        if m >= 2.0:
            raise err if err is not None else RuntimeError("another band failed in this stage")
        return m > 0.0

    def _cvote(self, err=None):
        """Vote on a rank-local failure on the CALLING thread's communicator before the next collective of that thread: either
        every rank goes on or every rank raises -- a rank that raised alone would leave the others waiting for it."""
        if self._cur_comm().allreduce_max(1.0 if err is not None else 0.0) > 0.0:
            raise err if err is not None else RuntimeError("another band failed in this stage")

    def _local(self, fn, *args):
        """run a band-local step, then vote on it (see _cvote)"""
        err, out = None, None
        try:
            out = fn(*args)
        except Exception as e:
            err = e
        self._cvote(err)
        return out

    def _swap_edges(self, name):
        """Neighbours trade edge rows of raster ``name``; returns which of my halo rows changed (top, bottom)."""
        b = self.band
        if self.rccl:
            return b.exchange_halo(name)
        if hasattr(b, "get_edge_rows"):      # (the HIP band: both rows per library call, two host synchronisations instead of four)
            to_up, to_down = b.get_edge_rows(name, self.has_up, self.has_down)
            from_up, from_down = self.comm.exchange_rows(to_up, to_down)
            return b.set_halo_rows(name, from_up if self.has_up else None, from_down if self.has_down else None)
        to_up = b.get_edge_row(name, 0) if self.has_up else None
        to_down = b.get_edge_row(name, 1) if self.has_down else None
        from_up, from_down = self.comm.exchange_rows(to_up, to_down)
        ch_top = b.set_halo_row(name, 0, from_up) if self.has_up else False
        ch_bot = b.set_halo_row(name, 1, from_down) if self.has_down else False
        return ch_top, ch_bot

    def _neighbour_rows(self, name):
        """(from_up, from_down): the neighbours' edge rows of raster ``name`` as host arrays, my halo rows untouched.  Over RCCL
        when the band owns a communicator for the calling thread, else through the thread's host communicator."""
        b = self.band
        on_side = getattr(self._tls, "comm", None) is not None
        if (self.rccl_side if on_side else self.rccl) and hasattr(b, "exchange_edge_rows"):
            return b.exchange_edge_rows(name)
        to_up = b.get_edge_row(name, 0) if self.has_up else None
        to_down = b.get_edge_row(name, 1) if self.has_down else None
        return self._cur_comm().exchange_rows(to_up, to_down)

    class _HostSection(object):
        """``with self._host("label"):`` charges the thread CPU time of a host-only section to ``host_ms[stage]`` (CPU time, not
        wall clock: several bands of one process take turns on the GIL in the one-GPU rehearsals)"""
        def __init__(self, owner, stage):
            self.owner, self.stage = owner, stage

        def __enter__(self):
            import time
            self.t0 = time.thread_time()

        def __exit__(self, *exc):
            import time
            self.owner.host_ms[self.stage] = self.owner.host_ms.get(self.stage, 0.0) + (time.thread_time() - self.t0) * 1e3
            return False

    def _host(self, stage):
        return BandPipeline._HostSection(self, stage)

    def upload_dem(self, dem_band):
        """``dem_band``: this rank's owned rows (nrows x W float32).  DEM halo rows are fetched from the neighbours."""
        self.band.upload("dem", dem_band)
        self._swap_edges("dem")

    def _fill(self, kind, name, short=0.0, diag=0.0, attach=False):
        b = self.band
        err, active = None, False
        lap = _Lap("fill loop %d" % kind, self.comm.rank)
        try:
            active = b.fill_attach(kind, short, diag) if attach else b.fill_begin(kind, short, diag)
        except Exception as e:
            err = e
        lap.lap("begin")
        while True:
            busy = False
            if err is None:
                try:
                    ch_top, ch_bot = self._swap_edges(name)
                    lap.lap("swap")
                    if ch_top:
                        b.fill_halo_changed(kind, 0)
                    if ch_bot:
                        b.fill_halo_changed(kind, 1)
                    self.exchanges["noflat" if kind else "fill"] += 1
                    busy = active or ch_top or ch_bot
                    lap.lap("halo_changed")
                except Exception as e:
                    err = e
            v = self._vote(busy, err)
            lap.lap("vote")
            if not v:
                # every band is quiescent: certify -- one sweep over EVERY tile of every band.  The worklist schedule only
                # revisits a tile when a neighbour's probe saw its halo drop; a sweep that moves nothing anywhere proves the
                # state is the global fixed point (and repairs a lost wake-up if there ever was one: the loop resumes)
                moved = False
                try:
                    moved = b.fill_certify(kind)
                except Exception as e:
                    err = e
                lap.lap("certify")
                if not self._vote(moved, err):
                    break
                active = False
                continue
            try:
                active = b.fill_batch(kind) if busy else False
            except Exception as e:
                err = e
            lap.lap("batch")
        b.fill_end(kind)
        lap.lap("end")
        lap.report()

    def fill(self):
        """fill.fill_terrain over all bands (+ bluespot depths of the owned rows)."""
        self.exchanges["fill"] = 0      # (per call, like the no-flats fill's: ``exchanges`` describes the last chain)
        self._fill(0, "filled")

    def short_and_diag(self):
        """fill.minimum_safe_short_and_diag on the GLOBAL raster (reference fill.py:235-250)."""
        mn, mx, has_nan = self.band.dem_minmax()
        parts = self.comm.allgather((float(mn), float(mx), bool(has_nan)))
        amax = np.float32(max(p[1] for p in parts))
        amin = np.float32(min(p[0] for p in parts))
        if any(p[2] for p in parts):
            # np.amax / np.amin propagate NaN (fill.py:246-249), and so does the one-context path (reduce.hip): a NaN cell in ANY
            # band makes short and diag NaN on every band
            amax = amin = np.float32("nan")
        maxval = np.float64(max(abs(amax), abs(amin)))
        nextval = np.nextafter(maxval, np.float64(float('inf')))
        short = (nextval - maxval) * 1024
        self.short, self.diag = float(short), float(short * (2 ** 0.5))
        return self.short, self.diag

    def noflat(self):
        """fill.fill_terrain_no_flats over all bands (needs fill())."""
        short, diag = self.short_and_diag()
        if hasattr(self.band, "geo_begin") and not (_DEV and os.environ.get("MALSTROEM_BAND_NOFLAT", "") == "relaxation") and self._noflat_geodesic(short, diag):
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
        self._local(self.band.run_flowdir)
        self._swap_edges("flowdir")

    def accum(self):
        """flow.accumulated_flow over all bands in two local passes, one neighbour exchange and one SMALL all-gather, however
        often the rivers cross the seams.  Pass 1 (``accum_boundary``) accumulates every band's OWN cells and finds, for each
        cell of a neighbour's edge row that flows into the band, the edge cell through which that flux leaves the band again
        (-1: it stays).  Neighbours trade the own-contribution edge rows (RCCL); only the PASS-THROUGH pairs
        (entering halo cell -> leaving edge cell, with their own contributions) are gathered: they form the forest
        x[e] = own[e] + sum(x[k] for the k that leave through e), solved identically on every rank (``mhip_band_forest_solve``); every
        other seam cell carries its own contribution.  Pass 2 is the local accumulation with the halo rows as known sources."""
        b, W, me = self.band, self.W, self.comm.rank
        self.exchanges["accum"] = 0

        def local_part():
            ex = b.accum_boundary()
            return ex, b.get_edge_row("accum", 0), b.get_edge_row("accum", 1)
        exit_map, own_first, own_last = self._local(local_part)
        nbr_up, nbr_dn = self._neighbour_rows("accum")
        self.exchanges["accum"] += 1
        with self._host("accum"):
            own_edge = np.ascontiguousarray(np.concatenate([own_first, own_last]), dtype=np.float64)
            exit_map = np.ascontiguousarray(exit_map, dtype=np.int32)
            # node (band, side, column) = (2 * band + side) * W + column; side 0 = first owned row, 1 = last owned row
            pairs = np.empty(2 * W, ACCUM_PAIR_DTYPE)
            top = None if nbr_up is None else np.array(nbr_up, dtype=np.float64)
            bot = None if nbr_dn is None else np.array(nbr_dn, dtype=np.float64)
            k = 0
            for half, nbr, base in ((exit_map[:W], top, (2 * (me - 1) + 1) * W), (exit_map[W:], bot, (2 * (me + 1)) * W)):
                if nbr is None:
                    continue
                n = ctypes.c_int64(0)
                _lib.call("mhip_band_accum_pairs", _lib.i64(W), _lib.ptr(half), _lib.ptr(nbr), _lib.ptr(own_edge), _lib.i64(base),
                          _lib.i64(2 * me * W), _lib.ptr(pairs[k:]), ctypes.byref(n))
                k += n.value
            mine = pairs[:k]
        parts = self._cur_comm().allgather(mine)
        with self._host("accum"):
            # (as bytes: NumPy concatenates structured arrays field by field, ten times slower)
            pairs = np.concatenate([np.ascontiguousarray(p, dtype=ACCUM_PAIR_DTYPE).view(np.uint8) for p in parts]).view(ACCUM_PAIR_DTYPE)
            if _PROF and me == 0:
                print("[band profile] accum: %d pass-through pairs of %d seam cells" % (pairs.size, 2 * W * (self.comm.size - 1)), file=sys.stderr,
                      flush=True)
            _lib.call("mhip_band_accum_solve", _lib.i64(2 * self.comm.size * W), _lib.i64(pairs.size), _lib.ptr(pairs), _lib.i64(W),
                      _lib.i64((2 * (me - 1) + 1) * W), _lib.ptr(top) if top is not None else None,
                      _lib.i64((2 * (me + 1)) * W), _lib.ptr(bot) if bot is not None else None)

        def second_pass():
            if self.has_up:
                b.set_halo_row("accum", 0, top)
            if self.has_down:
                b.set_halo_row("accum", 1, bot)
            b.run_accum()
        self._local(second_pass)

    def label(self, with_stats=False):
        """label.connected_components over all bands with scipy's numbering (order of first raster pixel).

        Every band labels its local raster (owned + halo rows).  Neighbours trade their edge rows of LOCAL labels (RCCL); a band
        compares each halo row (its own labels) with the neighbour's labels of the same raster row and publishes one
        (mine, theirs) pair per run -- a few thousand pairs where the rows hold 65536 cells -- together with its phantoms (local
        components made of halo cells only).  The pairs of all bands are gathered once; every rank finds the classes (union-find,
        ``mhip_band_union_find``) and derives the numbering of EVERY band from them, so no further exchange is needed; each band
        writes its labels on the device.  Returns the global number of labels.

        A band that can label in two halves (``ccl_begin`` / ``ccl_finish``: the HIP band) writes only its edge rows in band-local
        labels before the merge and every cell's GLOBAL label after it, in one pass over the raster instead of three (emit, read
        back + rewrite); ``with_stats``: the label statistics of ``stats()`` ride on that pass."""
        b, comm, W, R, me = self.band, self._cur_comm(), self.W, self.comm.size, self.comm.rank
        lap = _Lap("label", comm.rank)
        halves = hasattr(b, "ccl_begin") and not _LABEL_THREE_PASSES
        self._stats_on_device = False

        def local_part():
            n = b.ccl_begin() if halves else b.ccl_local()
            return n, dict(first=b.get_edge_row("labels", 0), last=b.get_edge_row("labels", 1),
                           top=b.get_edge_row("labels", 2) if self.has_up else None,
                           bot=b.get_edge_row("labels", 3) if self.has_down else None)
        nloc, rows = self._local(local_part)
        lap.lap("ccl_local + edge rows")
        started = getattr(self._tls, "ccl_done", None)      # run_chain: the main thread holds the accumulation back until here
        if started is not None:
            started.set()
        nbr_up, nbr_dn = self._neighbour_rows("labels")      # the neighbour's LAST row above / FIRST row below, in ITS labels
        lap.lap("neighbour rows")
        with self._host("label"):
            ea, eb, ph = np.empty(2 * W, np.int64), np.empty(2 * W, np.int64), np.empty(2 * W, np.int64)
            k = q = 0
            for halo, edge, nbr, r_nbr in ((rows["top"], rows["first"], nbr_up, me - 1), (rows["bot"], rows["last"], nbr_dn, me + 1)):
                if halo is None:
                    continue
                nbr = np.ascontiguousarray(nbr, dtype=np.int32)
                n, nq = ctypes.c_int64(0), ctypes.c_int64(0)
                # one pair per RUN of equal (mine, theirs) along the row; phantom = a local component without an owned cell
                _lib.call("mhip_band_label_pairs", _lib.i64(W), _lib.ptr(halo), _lib.ptr(edge), _lib.ptr(nbr), _lib.i64(me << 32), _lib.i64(r_nbr << 32),
                          _lib.ptr(ea[k:]), _lib.ptr(eb[k:]), ctypes.byref(n), _lib.ptr(ph[q:]), ctypes.byref(nq))
                k += n.value
                q += nq.value
            mine = dict(nloc=int(nloc), ea=ea[:k], eb=eb[:k], ph=ph[:q] | (np.int64(me) << 32))
        allp = comm.allgather(mine)
        lap.lap("allgather")
        with self._host("label"):
            EA = np.ascontiguousarray(np.concatenate([p["ea"] for p in allp]))
            EB = np.ascontiguousarray(np.concatenate([p["eb"] for p in allp]))
            PH = np.ascontiguousarray(np.concatenate([p["ph"] for p in allp]))
            if _PROF and me == 0:
                print("[band profile] label: %d seam pairs, %d phantoms" % (EA.size, PH.size), file=sys.stderr, flush=True)
            nlocs = np.array([p["nloc"] for p in allp], np.int64)
            cap = max(2 * EA.size + PH.size, 1)
            offsets = np.zeros(R + 1, np.int64)
            dropped, target, shared = np.empty(cap, np.int32), np.empty(cap, np.int32), np.empty(cap, np.int64)
            nd, ns = ctypes.c_int64(0), ctypes.c_int64(0)
            # the classes (phantoms may touch no foreground above / below: they are dropped labels all the same), the numbering of
            # every band, my dropped labels with their global labels, and the global labels with cells in more than one band
            _lib.call("mhip_band_label_merge", R, me, _lib.ptr(nlocs), _lib.i64(EA.size), _lib.ptr(EA), _lib.ptr(EB), _lib.i64(PH.size), _lib.ptr(PH),
                      _lib.ptr(offsets), _lib.ptr(dropped), _lib.ptr(target), ctypes.byref(nd), _lib.ptr(shared), ctypes.byref(ns))
            dropped, target = dropped[:nd.value], target[:nd.value]
            off = int(offsets[me])
            self.shared_labels = shared[:ns.value].copy()
            self.nlabels = int(offsets[-1])
            self.label_range = (int(offsets[me]) + 1, int(offsets[me + 1]))
            self.label_offsets = offsets
        lap.lap("merge (host)")
        if halves:
            self._local(b.ccl_finish, off, dropped.astype(np.int32), target.astype(np.int32), self.nlabels, with_stats)
            self._stats_on_device = bool(with_stats)
        else:
            self._local(b.relabel_sparse, nloc, off, dropped.astype(np.int32), target.astype(np.int32), self.nlabels)
        lap.lap("global labels")
        lap.report()
        return self.nlabels

    def watershed(self):
        """flow.watersheds_from_labels over all bands: local pointer jumping with pseudo labels on the halo rows (a cell whose path
        leaves the band through halo column k carries -(1 + k), bottom halo: -(1 + W + k)), then the pseudo labels are resolved
        through the neighbours' edge rows.  Neighbours trade their edge rows (RCCL): most paths end at a label right there.  What
        is left are paths that bounce back or cross a whole band; only the edge cells on such paths are gathered -- a cell is
        published when the neighbour points at it or when its own target is unresolved as well -- and every rank follows the
        chains to their end (``mhip_band_ws_resolve``; a flow cycle across bands stays unassigned like in _flow.pyx:276-314)."""
        b, comm, W, me = self.band, self._cur_comm(), self.W, self.comm.rank
        lap = _Lap("watershed", comm.rank)

        def local_part():
            b.watershed_local()
            return b.get_edge_row("watersheds", 0), b.get_edge_row("watersheds", 1)
        first, last = self._local(local_part)
        lap.lap("watershed_local + edge rows")
        nbr_up, nbr_dn = self._neighbour_rows("watersheds")      # the neighbour's LAST row above / FIRST row below
        lap.lap("neighbour rows")
        with self._host("watershed"):
            up = None if nbr_up is None else np.ascontiguousarray(nbr_up, dtype=np.int32)
            dn = None if nbr_dn is None else np.ascontiguousarray(nbr_dn, dtype=np.int32)
            mine = np.ascontiguousarray(np.concatenate([first, last]), dtype=np.int32)   # index e = side * W + column; node 2 me W + e
            N, V, n = np.empty(2 * W, np.int64), np.empty(2 * W, np.int64), ctypes.c_int64(0)
            pu = _lib.ptr(up) if up is not None else None
            pd = _lib.ptr(dn) if dn is not None else None
            _lib.call("mhip_band_ws_publish", _lib.i64(W), me, _lib.ptr(mine), pu, pd, _lib.ptr(N), _lib.ptr(V), ctypes.byref(n))
            mine_pub = (N[:n.value], V[:n.value])
        parts = comm.allgather(mine_pub)
        with self._host("watershed"):
            N = np.ascontiguousarray(np.concatenate([p[0] for p in parts]))
            V = np.ascontiguousarray(np.concatenate([p[1] for p in parts]))
            if _PROF and me == 0:
                print("[band profile] watershed: %d published chain cells" % N.size, file=sys.stderr, flush=True)
            lut = np.empty(2 * W, np.int32)
            _lib.call("mhip_band_ws_lut", _lib.i64(W), me, _lib.i64(N.size), _lib.ptr(N), _lib.ptr(V), pu, pd, _lib.ptr(lut))
        lap.lap("seam system (host + allgather)")
        self._local(b.apply_neg_lut, "watersheds", lut)
        lap.lap("apply_neg_lut")
        lap.report()

    # ---- per-label records (reference bluespots.py:159-206 on one raster).  Every rank returns the records of the labels
    # IT numbered (``label_range``, complete after the merge) plus the background record; a label or a watershed that
    # reaches into other bands is merged from their partial records (a handful of rows per exchange).
    def _cur_comm(self):
        """the communicator of the calling thread: the clone inside run_chain's labelling branch, else the main one"""
        return getattr(self._tls, "comm", None) or self.comm

    def _merged(self, which, fetch_own=True):
        """own-range slice of record set `which`, with the labels that live in several bands (and the background) merged in band
        order (``mhip_band_merge_records``: min / max / sums for the statistics; for the pour points the larger -- no accumulated
        flow: smaller -- value wins and the earlier band, i.e. the first raster position, on ties).  fetch_own=False: compute and
        merge only (the slice of this band's own labels stays on the device, like the records of the single-GPU pipeline until
        somebody asks for them)"""
        b, (lo, hi) = self.band, self.label_range
        lap = _Lap("records %d" % which, self.comm.rank)
        ids = np.concatenate([[0], self.shared_labels]).astype(np.int64)
        fresh = which == 0 and getattr(self, "_stats_on_device", False)      # (label(with_stats=True) left them on the device)
        g = self._local(lambda: (None if fresh else b.records_compute(which), b.records_gather(which, ids))[1])
        lap.lap("compute + gather %d" % ids.size)
        parts = self._cur_comm().allgather(g)
        lap.lap("allgather")
        with self._host("records"):
            parts = [np.ascontiguousarray(p, dtype=g.dtype) for p in parts]
            m = np.empty_like(g)
            pp = (ctypes.c_void_p * len(parts))(*[p.ctypes.data for p in parts])
            _lib.call("mhip_band_merge_records", int(which), len(parts), _lib.i64(g.size), pp, _lib.ptr(m))
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
        return self._merged(0, fetch_own)

    def watershed_counts(self, fetch_own=True):
        """label.label_count(watersheds): cells per watershed.  A watershed may reach into any band, so every rank publishes
        its non-zero counts of labels it did not number (a sparse handful) and adds what the others found of its own."""
        b, (lo, hi) = self.band, self.label_range
        def local_part():
            b.records_compute(1)
            own_ = b.records_fetch(1, lo, hi - lo + 1) if fetch_own else None
            fid_, fval_ = b.foreign_counts(lo, hi)
            return own_, fid_, fval_, int(b.records_fetch(1, 0, 1)[0])
        lap = _Lap("watershed_counts", self.comm.rank)
        own, fid, fval, bg = self._local(local_part)
        lap.lap("compute + fetch + foreign %d" % fid.size)
        lap.report()
        total0 = 0
        gathered = self._cur_comm().allgather((fid, fval, bg))
        with self._host("records"):
            for r, (ids, vals, c0) in enumerate(gathered):
                total0 += c0
                if r != self.comm.rank and own is not None:
                    mine = (ids >= lo) & (ids <= hi)
                    np.add.at(own, ids[mine] - lo, vals[mine])
        return {"first_label": lo, "records": own, "background": np.int64(total0)}

    def pourpoints(self, fetch_own=True, use_accum=True):
        """label.label_max_index(accum, labels) -- or label.label_min_index(no-flats surface, labels) when no accumulated flow
        was asked for (reference bluespots.py:195-206): value, row, col of the first raster cell with the extreme value per
        bluespot (rows are global)."""
        return self._merged(2 if use_accum else 3, fetch_own)

    # ---- the bluespot filter on bands (reference bluespots.py:23-46, 165-172)
    def filter(self, keep_of_records):
        """Keeps the bluespots for which ``keep_of_records(stats_records_of_my_labels) -> bool array`` holds and renumbers the rest
        1..n in raster order (the reference relabels the kept mask with a second connected_components run, which numbers the
        kept bluespots in the same order).  Every rank decides about the labels IT numbered (their records are complete after
        the merge), the kept counts give the new numbering, and a bluespot that reaches into other bands is looked up in a
        short published list.  Returns the new global number of labels."""
        b, comm, me = self.band, self._cur_comm(), self.comm.rank
        lo, hi = self.label_range
        st = self.stats(fetch_own=True)
        with self._host("filter"):
            keep = np.asarray(keep_of_records(st["records"]), dtype=bool)
            if keep.shape != (hi - lo + 1,):
                raise ValueError("the filter must return one boolean per record")
            nkept = int(keep.sum())
        counts = comm.allgather(nkept)
        with self._host("filter"):
            off = int(np.sum(counts[:me]))
            lut = np.where(keep, off + np.cumsum(keep), 0).astype(np.int64)            # new label of my label lo + i
            sh = self.shared_labels
            mine_sh = sh[(sh >= lo) & (sh <= hi)]
            pub = (mine_sh.astype(np.int64), lut[mine_sh - lo].astype(np.int64))
        published = comm.allgather(pub)
        with self._host("filter"):
            ids = np.concatenate([p[0] for p in published])
            new = np.concatenate([p[1] for p in published])
            order = np.argsort(ids)
            ids, new = ids[order], new[order]
            foreign = (ids < lo) | (ids > hi)
            self.nlabels = int(np.sum(counts))
            self.label_range = (off + 1, off + nkept)
            self.shared_labels = np.unique(new[new > 0]).astype(np.int64)
        self._local(b.relabel_range, lo, hi, lut.astype(np.int32), ids[foreign].astype(np.int32), new[foreign].astype(np.int32), self.nlabels)
        self._stats_on_device = False
        return self.nlabels

    # ---- the stream walk across bands (reference net.py:142-169; one leg per band: csrc/trace.hip band_trace_kernel)
    def trace_downstream(self, cells, background_label=None, geometry=False):
        """``next_downstream_label`` for many cells of the GLOBAL raster.  Every rank passes the same ``cells`` (global (row, col),
        e.g. the merged pour points); a walker is started by the band that owns its cell, walks while it stays on that band's
        rows and is handed to the neighbour when it steps across a seam (one small all-gather per hop: walker id, source label,
        cell), until no walker is left.  Returns ``(labels, geoms)`` like ``algorithms.net.trace_downstream_labels`` -- the same
        on every rank: ``labels[i]`` int or None, ``geoms[i]`` a list of (row, col) (empty unless ``geometry``)."""
        b, comm, W, me = self.band, self._cur_comm(), self.W, self.comm.rank
        cells = np.asarray(list(cells), dtype=np.int64).reshape(-1, 2)
        n = cells.shape[0]
        owned = lambda rows: (rows >= self.row0) & (rows < self.row0 + self.nrows)
        ids = np.flatnonzero(owned(cells[:, 0]))
        cur_cells, cur_src = cells[ids], np.full(ids.size, -1, np.int32)
        results = {}           # walker id -> (label or None)
        legs = []              # (walker id, hop number, cells of that leg)
        hop = 0
        while True:
            def leg():
                return b.trace(cur_cells, cur_src, background_label, geometry) if ids.size else (np.zeros(0, np.int32),) * 3 + (np.zeros((0, 2), np.int64), [])
            lab, status, src_out, exits, geoms = self._local(leg)
            for k, wid in enumerate(ids.tolist()):
                if geometry:
                    legs.append((wid, hop, geoms[k]))
                if status[k] == 1:
                    results[wid] = int(lab[k])
                elif status[k] == 0:
                    results[wid] = None
            go = status == 2
            handed = comm.allgather((ids[go].astype(np.int64), src_out[go].astype(np.int32), exits[go].astype(np.int64)))
            all_ids = np.concatenate([h[0] for h in handed])
            if all_ids.size == 0:
                break
            all_src = np.concatenate([h[1] for h in handed])
            all_ex = np.concatenate([h[2] for h in handed]).reshape(-1, 2)
            take = owned(all_ex[:, 0])
            ids, cur_src, cur_cells = all_ids[take], all_src[take], all_ex[take]
            hop += 1
            if hop > self.H * self.W:       # (a walker bouncing forever between two bands: a flow cycle across a seam)
                for wid in all_ids.tolist():
                    results.setdefault(wid, None)
                break
        merged = comm.allgather((results, legs if geometry else []))
        labels = [None] * n
        parts = {}
        for res, lg in merged:
            for wid, l in res.items():
                labels[int(wid)] = l
            for wid, h, cells_ in lg:
                parts.setdefault(int(wid), []).append((int(h), cells_))
        geoms = [[] for _ in range(n)]
        if geometry:
            for wid, pieces in parts.items():
                flat = np.concatenate([c for _, c in sorted(pieces, key=lambda t: t[0])]) if pieces else np.zeros(0, np.int64)
                rows, cols = np.divmod(flat, W)
                geoms[wid] = list(zip(rows.tolist(), cols.tolist()))
        return labels, geoms

    def gather_rows(self, name, max_rows=1024):
        """The raster ``name`` of the whole DEM, band after band in pieces of ``max_rows`` rows: yields (row0, rows) on rank 0 and
        (row0, None) on the other ranks (every rank iterates in step: one collective per piece)."""
        comm = self._cur_comm()
        extents = comm.allgather((self.row0, self.nrows))
        mine = self.band.download(name) if self.nrows else None
        for r, (r0, nr) in enumerate(extents):
            for a in range(0, nr, int(max_rows)):
                piece = mine[a:a + int(max_rows)] if r == self.comm.rank else None
                got = comm.allgather(piece)[r]
                yield r0 + a, (got if self.comm.rank == 0 else None)

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
        err, err_side = [], []      # failures of the main thread / of the labelling thread
        ccl_done = threading.Event()     # the labelling's tile pass is through (or the labelling has failed)

        def label_branch():
            try:
                if hasattr(self.band, "side_begin"):
                    self.band.side_begin()
                self._tls.comm = self._comm_b
                self._tls.ccl_done = ccl_done
                timed("label", lambda: self.label(with_stats=records))
                if records:
                    out["stats"] = timed("label", lambda: self.stats(fetch_own))
            except Exception as e:      # re-raised on the main thread
                err_side.append(e)
            finally:
                ccl_done.set()
                self._tls.ccl_done = None
                self._tls.comm = None
                if hasattr(self.band, "side_end"):
                    try:
                        self.band.side_end()
                    except Exception as e:
                        err_side.append(e)

        def ws_branch():
            try:
                if hasattr(self.band, "side_begin"):
                    self.band.side_begin()
                self._tls.comm = self._comm_b
                timed("watershed", self.watershed)
                if records:
                    out["counts"] = timed("watershed", lambda: self.watershed_counts(fetch_own))
            except Exception as e:
                err_side.append(e)
            finally:
                self._tls.comm = None
                if hasattr(self.band, "side_end"):
                    try:
                        self.band.side_end()
                    except Exception as e:
                        err_side.append(e)

        if not overlap:
            timed("noflat", self.noflat)
            timed("flowdir", self.flowdir)
            timed("accum", self.accum)
            timed("label", lambda: self.label(with_stats=records))
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
        flowdir_failed = []

        def together(main_steps, side):
            """main_steps on this thread next to `side` on a second one.  Every stage votes on its rank-local failures before its
            next collective (_cvote / _vote), so a failure surfaces on EVERY rank at the same point of the same thread; the vote
            after the join makes all ranks raise together whichever thread it was."""
            th = threading.Thread(target=side)
            th.start()
            try:
                for name, fn in main_steps:
                    timed(name, fn)
            except Exception as e:
                err.append(e)
            finally:
                th.join()
            if self.comm.allreduce_max(1.0 if (err or err_side) else 0.0) > 0.0:
                raise (err + err_side)[0] if (err or err_side) else RuntimeError("another band failed in this phase of the chain")

        try:
            # like the single-GPU DAG (csrc/api.hip, measured there): the no-flats fill has the GPU to itself -- its many small
            # launches queue behind the labelling's long workgroups otherwise -- then labelling + watersheds (the latter
            # need labels and flow directions) run next to D8 + accumulation.  (Round 4: the labelling beside the no-flats fill, as in
            # one context: the no-flats stage 29 -> 43 ms at 4 bands of 32768^2 on one GPU and the step 115.7 / 119.2 -> 114.7 / 118.5
            # -- the band step is bound by the device's throughput, not by the order of its work.)
            timed("noflat", self.noflat)
            if self.comm.allreduce_max(0.0) > 0.0:      # (keeps the bands in step before the two-thread phase)
                raise RuntimeError("another band failed in the no-flats fill")
            flow_ready = threading.Event()

            def flowdir_then_signal():
                try:
                    self.flowdir()
                except Exception:
                    flowdir_failed.append(True)     # (the same on every rank: flowdir() votes before it raises)
                    raise
                finally:
                    flow_ready.set()

            labels_ready = threading.Event()

            def side():
                label_branch()
                labels_ready.set()
                flow_ready.wait()
                # whether the watershed branch runs is a COLLECTIVE decision on the side communicator: the facts it rests on
                # (label failed, flowdir failed) are the same on every rank, the vote makes sure of it
                bad = 1.0 if (err_side or flowdir_failed) else 0.0
                if self._comm_b.allreduce_max(bad) == 0.0:
                    ws_branch()

            def pour_next_to_the_watersheds():
                # the pour points want the accumulated flow (this thread) and the labels (the other one): they run here, on the main
                # stream and the main communicator, while the side thread is at the watersheds
                labels_ready.wait()
                if self.comm.allreduce_max(1.0 if err_side else 0.0) == 0.0:      # (label() failed: the same on every rank; the join below raises)
                    out["pour"] = self.pourpoints(fetch_own)

            def accum_behind_the_tile_labelling():
                # labelling -> statistics -> watersheds -> counts is the longer of the two branches, and its first kernel (the tile
                # union-find) took 19 ms instead of 6 next to the accumulation's tile pass (4 bands of 32768^2 on one GPU: the
                # device divides itself evenly, not by who is on the critical path); the accumulation has the slack to wait
                if not _ACCUM_AT_ONCE:
                    ccl_done.wait()
                self.accum()

            together([("flowdir", flowdir_then_signal), ("accum", accum_behind_the_tile_labelling)]
                     + ([("pourpoints", pour_next_to_the_watersheds)] if records else []), side)
        finally:
            sys.setswitchinterval(old_switch)
        return out

    def download(self, name):
        return self.band.download(name)

    def download_rows(self, name, row0, nrows):
        if hasattr(self.band, "download_rows"):
            return self.band.download_rows(name, row0, nrows)
        return self.band.download(name)[row0:row0 + nrows]

    def write_raster(self, name, writer):
        """Raster ``name`` of all bands into ONE file, every rank writing the tile rows that start in its band
        (``malstroem_amd.io.BandRasterWriter``): no rank gathers the raster, a rank's host memory is one tile row.  Collective."""
        extents = [tuple(e) for e in self._cur_comm().allgather((int(self.row0), int(self.nrows)))]
        return writer.write(self._cur_comm(), (self.H, self.W), extents, lambda r0, n: self.download_rows(name, r0, n), RASTER_DTYPE[RASTERS[name]])
