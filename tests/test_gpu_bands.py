"""GPU: several row bands of one DEM (one band context each, driven by threads over the in-process ThreadComm on a
single MI355X) must reproduce the undivided raster bit for bit: fill, depths, no-flats fill, D8, accumulation,
bluespot labels (scipy numbering) and watersheds."""
import threading

import numpy as np
import pytest

import oracle
from _cases import fbm, fixtures, meander_flowdir, random_flowdir, serpentine_flowdir, zigzag_flowdir

pytestmark = pytest.mark.gpu


def run_bands(dem, nbands, align=True):
    from malstroem_amd.distributed import BandPipeline, ThreadComm
    out = [None] * nbands
    err = []

    def work(comm):
        try:
            p = BandPipeline(comm, dem.shape, device=0, align=align)
            p.upload_dem(dem[p.row0:p.row0 + p.nrows])
            rec = p.run_chain(overlap=comm.size != 3)      # two host threads per band (3 bands: one after the other)
            n = p.nlabels
            out[comm.rank] = {k: p.download(k) for k in ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds")}
            out[comm.rank]["nlabels"] = n
            out[comm.rank]["short_diag"] = (p.short, p.diag)
            out[comm.rank]["exchanges"] = dict(p.exchanges)
            out[comm.rank]["engines"] = (p.band.get_int("fill_algorithm"), p.band.get_int("noflat_algorithm"))
            out[comm.rank].update(rec)
            p.close()
        except Exception as e:  # pragma: no cover
            err.append(e)
            raise

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(600) for t in threads]
    assert not err, err
    return out


@pytest.mark.parametrize("nbands", [2, 3])
def test_bands_match_reference_fixture(nbands):
    fx = fixtures()
    out = run_bands(fx["dtm"], nbands)
    assert np.array_equal(np.concatenate([o["filled"] for o in out]), fx["filled"])
    assert np.array_equal(np.concatenate([o["depths"] for o in out]), fx["depths"])
    assert np.array_equal(np.concatenate([o["noflat"] for o in out]), fx["filled_no_flats"])
    assert np.array_equal(np.concatenate([o["flowdir"] for o in out]), fx["flowdir_noflats"])
    assert out[0]["short_diag"] == (7.275957614183426e-12, 1.0289757937229989e-11)
    acc = np.concatenate([o["accum"] for o in out])
    assert acc.min() >= 1 and acc.max() == 11158 and acc.sum() == 3578615
    lab, n = oracle.connected_components(fx["depths"])
    assert out[0]["nlabels"] == n == 523 and np.array_equal(np.concatenate([o["labels"] for o in out]), lab)
    ws = lab.copy()
    oracle.watersheds_from_labels(fx["flowdir_noflats"], ws, 0)
    assert np.array_equal(np.concatenate([o["watersheds"] for o in out]), ws)


@pytest.mark.parametrize("nbands,h,w", [(2, 700, 450), (4, 1024, 1024), (4, 1000, 130), (3, 380, 130), (2, 252, 190), (5, 640, 70),
                                        (2, 128, 250), (3, 189, 127)])
def test_bands_match_oracle(nbands, h, w):
    dem = fbm(h, w, beta=2.0, seed=21)
    out = run_bands(dem, nbands)
    check_bands_against_oracle(dem, out)
    # seams on the tile grid: every band ran the tiled priority-flood and the geodesic no-flats transform, a handful of exchanges
    assert all(o["engines"] == (1, 2) for o in out), [o["engines"] for o in out]
    assert out[0]["exchanges"]["fill"] <= 12 and out[0]["exchanges"]["noflat"] <= 12


@pytest.mark.parametrize("nbands,h,w", [(2, 700, 450), (4, 1000, 130), (3, 380, 260)])
def test_bands_with_self_listing_tail_rounds(monkeypatch, nbands, h, w):
    """the geodesic rounds that build their own lists (the tail at 16384^2; MHIP_NG_BATCH=1: every round after the first here) across
    halo exchanges: an exchange wakes tiles through the mark bytes, the next round compacts them, the rounds after it append again"""
    monkeypatch.setenv("MHIP_NG_BATCH", "1")
    dem = fbm(h, w, beta=2.0, seed=24)
    dem[h // 2 - 30:h // 2 + 30, w // 5:w // 5 + 80] = dem[h // 2, w // 5]           # a flat across a seam
    out = run_bands(dem, nbands)
    check_bands_against_oracle(dem, out)
    assert all(o["engines"][1] == 2 for o in out), [o["engines"] for o in out]


@pytest.mark.parametrize("nbands,h,w", [(2, 700, 450), (3, 380, 130), (4, 1000, 130)])
def test_bands_off_the_tile_grid_run_the_iterative_fill(nbands, h, w):
    dem = fbm(h, w, beta=2.0, seed=22)
    out = run_bands(dem, nbands, align=False)
    check_bands_against_oracle(dem, out)
    assert all(o["engines"][0] == 0 for o in out[:-1])      # (the last band has no bottom halo row: the flood runs there)


@pytest.mark.parametrize("case", ["sea at elevation 0", "a flat beyond the uint32 headroom"])
def test_bands_attach_the_relaxation_to_a_partial_geodesic_surface(case):
    if case.startswith("sea"):
        dem = fbm(700, 450, beta=2.0, seed=23) - np.float32(30.0)
        dem[dem < 0] = 0.0
        nbands = 3
    else:       # the snake channel of test_gpu_noflat_geodesic.py, cut into bands
        h, w = 403, 203
        dem = np.full((h, w), 100.0, np.float32)
        for k, r in enumerate(range(1, h - 1, 2)):
            dem[r, 1:w - 1] = 0.3
            if r + 2 < h - 1:
                dem[r + 1, (w - 2) if k % 2 == 0 else 1] = 0.3
        dem[1, 0] = 0.2
        nbands = 2
    out = run_bands(dem, nbands)
    check_bands_against_oracle(dem, out)
    assert all(o["engines"][1] == 0 for o in out)        # the relaxation had the last word on the no-flats surface ...
    assert out[0]["exchanges"]["noflat"] >= 2            # ... after the geodesic exchanges


def check_bands_against_oracle(dem, out):
    filled = oracle.fill_terrain(dem)
    short, diag = oracle.minimum_safe_short_and_diag(dem)
    fnf = oracle.fill_terrain_no_flats(dem, short, diag)
    assert np.array_equal(np.concatenate([o["filled"] for o in out]), filled)
    assert np.array_equal(np.concatenate([o["noflat"] for o in out]), fnf)
    fd = oracle.terrain_flowdirection(fnf)
    assert np.array_equal(np.concatenate([o["flowdir"] for o in out]), fd)
    assert out[0]["short_diag"] == (short, diag)
    assert np.array_equal(np.concatenate([o["accum"] for o in out]), oracle.accumulated_flow(fd))
    lab, n = oracle.connected_components(oracle.depths(filled, dem))
    assert all(o["nlabels"] == n for o in out)
    assert np.array_equal(np.concatenate([o["labels"] for o in out]), lab)
    ws = lab.copy()
    oracle.watersheds_from_labels(fd, ws, 0)
    assert np.array_equal(np.concatenate([o["watersheds"] for o in out]), ws)
    # per-label records: every rank returns the labels it numbered, merged across bands
    acc = np.concatenate([o["accum"] for o in out])
    dep = oracle.depths(filled, dem)
    want = {"stats": oracle.label_stats(dep, lab, n), "counts": np.bincount(ws.ravel(), minlength=n + 1),
            "pour": oracle.label_max_index(acc, lab, n)}
    for key in ("stats", "counts", "pour"):
        got = np.concatenate([o[key]["records"] for o in out])
        assert len(got) == n
        w_ = want[key][1:]
        if key == "counts":
            assert np.array_equal(got, w_)
            assert all(int(o[key]["background"]) == int(want[key][0]) for o in out)
        elif key == "stats":
            for f in ("min", "max", "count"):
                assert np.array_equal(got[f], w_[f]), f
            assert np.allclose(got["sum"], w_["sum"], rtol=1e-12, atol=0)
        else:
            for f in ("value", "row", "col"):
                assert np.array_equal(got[f], w_[f]), f


def band_accum(fd, nbands):
    """BandPipeline.accum() on given flow directions -> (accumulation of the whole raster, exchanges per band)"""
    from malstroem_amd.distributed import BandPipeline, ThreadComm
    out, err = [None] * nbands, []

    def work(comm):
        try:
            p = BandPipeline(comm, fd.shape, device=0)
            p.band.upload("flowdir", fd[p.row0:p.row0 + p.nrows])
            p._swap_edges("flowdir")
            p.accum()
            out[comm.rank] = (p.download("accum"), p.exchanges["accum"], p.band.get_int("accum_algorithm"))
            p.close()
        except Exception as e:  # pragma: no cover
            err.append(e)
            raise

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(600) for t in threads]
    assert not err, err
    band_accum.algorithms = [o[2] for o in out]      # 1: the second pass ran as a delta over the boundary pass's perimeter graph
    return np.concatenate([o[0] for o in out]), [o[1] for o in out]


# band heights put the bottom halo row on the first row of a 64-row tile (64- and 63-row bands), in the middle of one, and
# make bands of a single row (first owned row == last owned row)
@pytest.mark.parametrize("name,fd,nbands", [
    ("serpentine", serpentine_flowdir(300, 200), 3), ("serpentine-64", serpentine_flowdir(192, 130), 3),
    ("serpentine-1row-bands", serpentine_flowdir(6, 300), 6), ("random", random_flowdir(260, 333, 1), 4),
    ("random-63", random_flowdir(189, 257, 2), 3), ("random-64", random_flowdir(128, 64, 3), 2),
    ("random-1row-bands", random_flowdir(5, 700, 4), 5), ("random-sparse", random_flowdir(500, 190, 5, p_none=0.5), 6),
    ("meander", meander_flowdir(1000, 700, 6), 4), ("meander-big", meander_flowdir(2048, 2048, 7), 4),
    ("random-big", random_flowdir(2048, 1500, 8), 5), ("zigzag", zigzag_flowdir(1000, 700, 12), 4), ("zigzag-big", zigzag_flowdir(2048, 1536, 13), 5),
    ("zigzag-thin-bands", zigzag_flowdir(40, 900, 14), 8)])
def test_band_accumulation_needs_one_exchange_however_often_the_flow_crosses_the_seams(name, fd, nbands):
    acc, exchanges = band_accum(fd, nbands)
    want = oracle.accumulated_flow(fd)
    bad = np.argwhere(acc != want)
    assert bad.size == 0, (name, len(bad), bad[:5].tolist(), acc[tuple(bad[0])], want[tuple(bad[0])])
    assert exchanges == [1] * nbands
    if name.startswith("serpentine"):
        assert acc.max() == fd.size      # the river collects every cell
    if name.startswith(("serpentine", "zigzag")):
        # acyclic flow: every band adds the flux that enters at its seams along the kept graph (accum.hip: accum_band_delta_dev);
        # an unknown halo value (a flow cycle upstream in the neighbouring band: the random and the meander grids hold two-cell
        # cycles) sends a band to the full pass
        assert band_accum.algorithms == [1] * nbands, band_accum.algorithms


def test_device_row_entry_points_match_host_ones():
    """mhip_ctx_get_edge_row_dev / set_halo_row_dev (what the RCCL transport calls) against the host-buffer variants.
    The device buffer comes straight from hipMalloc of the HIP runtime the library already uses (importing torch.cuda
    AFTER libmalstroem_hip would bring a second HIP runtime into the process)."""
    import ctypes
    from malstroem_amd.distributed import HipBand
    dem = fbm(96, 80, beta=2.0, seed=5)
    top = HipBand(96, 80, 0, 48, device=0, rank=0, size=2)
    bot = HipBand(96, 80, 48, 48, device=0, rank=1, size=2)
    hip = ctypes.CDLL("libamdhip64.so.7")
    buf = ctypes.c_void_p()
    nbytes = top.row_bytes("dem")
    assert hip.hipMalloc(ctypes.byref(buf), ctypes.c_size_t(nbytes)) == 0
    try:
        top.upload("dem", dem[:48])
        bot.upload("dem", dem[48:])
        top.get_edge_row_dev("dem", 1, buf.value)          # last owned row of the upper band
        host = np.empty(80, np.float32)
        assert hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), buf, ctypes.c_size_t(nbytes), 2) == 0   # 2 = device to host
        assert np.array_equal(host, dem[47])
        assert np.array_equal(top.get_edge_row("dem", 1), dem[47])
        bot.set_halo_row_dev("dem", 0, buf.value)
        assert np.array_equal(bot.get_edge_row("dem", 2), dem[47])
        assert bot.set_halo_row_dev("dem", 0, buf.value) is False      # same bytes again: unchanged
    finally:
        hip.hipFree(buf)
        top.close()
        bot.close()


def test_rccl_inside_the_library_single_rank():
    """The library's own RCCL transport (comm.hip: librccl opened at run time, ncclGetUniqueId, ncclCommInitRank, the
    all-reduce that ends the fill loops, ncclCommDestroy) on the only world size a one-GPU box offers; the band chain runs
    through BandPipeline's RCCL code path.  Neighbour traffic (ncclSend / ncclRecv between two GPUs) first runs on the
    driver's multi-GPU node; its protocol is covered by tests/test_distributed_cpu.py on the socket stand-in."""
    from malstroem_amd.distributed import BandPipeline, HipBand, SingleComm
    uid = HipBand.new_unique_id()
    assert len(uid) == 128 and any(uid)
    band = HipBand(64, 48, 0, 64, device=0, rank=0, size=1, unique_id=uid)
    try:
        assert band.has_comm
        assert band.allreduce_max(3.5) == 3.5 and band.allreduce_max(-2.0) == -2.0
        band.upload("dem", fbm(64, 48, seed=3))
        assert band.exchange_halo("dem") == (False, False)      # no neighbours: nothing moves
        assert band.exchange_edge_rows("dem") == (None, None)
        assert HipBand.comm_available()
        band.add_side_comm(HipBand.new_unique_id())             # the second communicator (ncclCommInitRank again on the same device)
        assert band.has_side_comm
        band.side_begin()
        assert band.exchange_edge_rows("dem") == (None, None)   # ... which the thread inside the side bracket uses
        band.side_end()
    finally:
        band.close()

    class OneRankRccl(SingleComm):
        size = 1

    dem = fbm(200, 150, beta=2.0, seed=9)
    p = BandPipeline(OneRankRccl(), dem.shape, device=0, rccl=False)
    p.band.close()
    p.band = HipBand(200, 150, 0, 200, device=0, rank=0, size=1, unique_id=HipBand.new_unique_id())
    p.band.add_side_comm(HipBand.new_unique_id())
    p.rccl, p.rccl_side = True, True
    try:
        p.upload_dem(dem)
        out = p.run_chain()
        got = {k: p.download(k) for k in ("filled", "noflat", "flowdir", "accum", "labels", "watersheds")}
        n = p.nlabels
    finally:
        p.close()
    filled = oracle.fill_terrain(dem)
    short, diag = oracle.minimum_safe_short_and_diag(dem)
    fnf = oracle.fill_terrain_no_flats(dem, short, diag)
    fd = oracle.terrain_flowdirection(fnf)
    assert np.array_equal(got["filled"], filled) and np.array_equal(got["noflat"], fnf) and np.array_equal(got["flowdir"], fd)
    assert np.array_equal(got["accum"], oracle.accumulated_flow(fd))
    lab, nref = oracle.connected_components(oracle.depths(filled, dem))
    assert n == nref and np.array_equal(got["labels"], lab)
    assert out["pour"] is not None


@pytest.mark.parametrize("nbands,h,w", [(2, 700, 450), (3, 640, 390)])
def test_band_flood_is_proven_and_repaired_across_bands(monkeypatch, nbands, h, w):
    """The flood of every band is proven when the bands are quiescent (mhip_ctx_fill_certify: check.hip).  MHIP_PF_CORRUPT raises one
    cell in the middle of every band afterwards: each band has to notice, continue on the iterative schedule from its surface
    (fill_algorithm 4), trade edge rows again until everything is quiet -- and the chain still matches the oracle bit for bit."""
    monkeypatch.setenv("MHIP_PF_CORRUPT", "1")
    dem = fbm(h, w, beta=2.0, seed=31)
    out = run_bands(dem, nbands)
    check_bands_against_oracle(dem, out)
    assert all(o["engines"][0] == 4 for o in out), [o["engines"] for o in out]


@pytest.mark.parametrize("name,h,w,nbands", [("serpentine", 300, 200, 3), ("serpentine-1row-bands", 6, 300, 6), ("zigzag", 1000, 700, 4),
                                            ("zigzag-thin-bands", 40, 900, 8), ("zigzag-big", 2048, 1024, 5)])
def test_band_watersheds_resolve_paths_that_bounce_and_cross_whole_bands(name, h, w, nbands):
    """the watershed seam protocol (neighbour rows + published chain cells) on flow that wanders across the seams"""
    from malstroem_amd.distributed import BandPipeline, ThreadComm
    fd = serpentine_flowdir(h, w) if name.startswith("serpentine") else zigzag_flowdir(h, w, 11)
    fd = fd.copy()
    fd[0], fd[-1], fd[:, 0], fd[:, -1] = 0, 4, 6, 2          # edges flow outward (flow.py:118-139)
    fd[0, 0], fd[0, -1], fd[-1, 0], fd[-1, -1] = 7, 1, 5, 3
    rng = np.random.default_rng(5)
    lab = np.zeros((h, w), np.int32)
    for k in range(1, 9):
        r, c = int(rng.integers(1, h - 1)), int(rng.integers(1, w - 3))
        lab[r:r + 2, c:c + 3] = k
    want = lab.copy()
    oracle.watersheds_from_labels(fd, want, 0)
    out = [None] * nbands

    def work(comm):
        p = BandPipeline(comm, fd.shape, device=0)
        p.band.upload("flowdir", fd[p.row0:p.row0 + p.nrows])
        p._swap_edges("flowdir")
        p.band.upload("labels", lab[p.row0:p.row0 + p.nrows])
        p._swap_edges("labels")
        p.watershed()
        out[comm.rank] = p.download("watersheds")
        p.close()

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(600) for t in threads]
    assert all(o is not None for o in out)
    assert np.array_equal(np.concatenate(out), want), name


def test_bands_with_a_nan_cell_follow_the_one_context_path():
    """A NaN cell makes np.amax / np.amin -- and with them short and diag -- NaN (fill.py:246-249); the one-context path propagates it
    (reduce.hip) and so do the bands now (the band path used to take a finite epsilon).  The plain fill ignores the NaN
    cell like the reference (it never moves, never wins a minimum): the bands equal the undivided raster.  (With NaN epsilons the
    no-flats fill is undefined in the reference -- its two variants disagree -- and does not converge here on either path.)"""
    import malstroem_amd.algorithms as alg
    from malstroem_amd.distributed import BandPipeline, ThreadComm
    dem = fbm(300, 260, beta=2.0, seed=41)
    dem[200, 100] = np.nan
    sd = alg.fill.minimum_safe_short_and_diag(dem)
    assert np.isnan(sd[0]) and np.isnan(sd[1])
    want = alg.fill.fill_terrain(dem)
    out = [None] * 3

    def work(comm):
        p = BandPipeline(comm, dem.shape, device=0)
        p.upload_dem(dem[p.row0:p.row0 + p.nrows])
        p.fill()
        out[comm.rank] = (p.download("filled"), p.short_and_diag())
        p.close()

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(3)]
    [t.start() for t in threads]
    [t.join(600) for t in threads]
    assert all(o is not None for o in out)
    assert all(np.isnan(o[1][0]) and np.isnan(o[1][1]) for o in out)
    assert np.array_equal(np.concatenate([o[0] for o in out]), want, equal_nan=True)


@pytest.mark.parametrize("nbands,h,w,with_stats", [(3, 700, 450, True), (4, 1030, 300, False), (2, 5, 300, True), (5, 640, 70, True)])
def test_labelling_in_two_halves_equals_the_three_passes(monkeypatch, nbands, h, w, with_stats):
    """ccl_begin / ccl_finish (edge rows before the seam merge, every cell's GLOBAL label -- and the label statistics -- in one pass
    after it) against ccl_local + relabel_sparse + records_compute(0): the same labels on owned AND halo rows, the same records"""
    import malstroem_amd.distributed as D
    dem = fbm(h, w, beta=2.0, seed=31)
    dem[h // 2 - 2:h // 2 + 2, 10:w - 10] = dem.min() - 1.0          # a bluespot along (and across) the middle seams
    filled = oracle.fill_terrain(dem)
    dep = oracle.depths(filled, dem)

    def run(three_passes):
        monkeypatch.setattr(D, "_LABEL_THREE_PASSES", three_passes)
        out, err = [None] * nbands, []

        def work(comm):
            try:
                p = D.BandPipeline(comm, dem.shape, device=0)
                p.upload_dem(dem[p.row0:p.row0 + p.nrows])
                p.fill()
                n = p.label(with_stats=with_stats)
                assert p._stats_on_device == (with_stats and not three_passes)
                st = p.stats(fetch_own=True)
                rows = [p.band.get_edge_row("labels", side) for side in (0, 1)]
                rows += [p.band.get_edge_row("labels", 2) if p.has_up else None, p.band.get_edge_row("labels", 3) if p.has_down else None]
                out[comm.rank] = (n, p.download("labels"), st, rows)
                p.close()
            except Exception as e:  # pragma: no cover
                err.append(e)
                raise
        threads = [threading.Thread(target=work, args=(c,)) for c in D.ThreadComm.world(nbands)]
        [t.start() for t in threads]
        [t.join(600) for t in threads]
        assert not err, err
        return out

    a, b = run(False), run(True)
    lab, n = oracle.connected_components(dep)
    assert np.array_equal(np.concatenate([o[1] for o in a]), lab) and all(o[0] == n for o in a)
    want = oracle.label_stats(dep, lab, n)
    got = np.concatenate([o[2]["records"] for o in a])
    for f in ("min", "max", "count"):
        assert np.array_equal(got[f], want[1:][f]), f
    for x, y in zip(a, b):
        assert x[0] == y[0] and np.array_equal(x[1], y[1])
        assert np.array_equal(x[2]["records"], y[2]["records"]) and x[2]["background"] == y[2]["background"]
        for r, q in zip(x[3], y[3]):
            assert (r is None and q is None) or np.array_equal(r, q)
    # the halo rows carry the neighbour's global labels
    for k in range(1, nbands):
        assert np.array_equal(a[k][3][2], a[k - 1][3][1]) and np.array_equal(a[k - 1][3][3], a[k][3][0])


@pytest.mark.parametrize("nbands", [2, 3])
def test_bands_of_a_terrain_without_bluespots(nbands):
    """no depression anywhere (a tilted plane with one pit in the LAST band only / none at all): bands without a single label go through
    the labelling's two halves, the statistics pass and the record merges with empty label ranges"""
    h, w = 400, 300
    plane = (np.add.outer(np.arange(h), np.arange(w)) * 0.25).astype(np.float32)
    for pit in (False, True):
        dem = plane.copy()
        if pit:
            dem[h - 20:h - 10, 50:60] -= 30.0
        out = run_bands(dem, nbands)
        check_bands_against_oracle(dem, out)
        assert out[0]["nlabels"] == (1 if pit else 0)


def test_both_rows_of_an_exchange_in_one_call():
    """mhip_ctx_get_edge_rows / mhip_ctx_set_halo_rows (one synchronisation per call: what the host transport's halo exchange uses)
    against the one-row calls, on a middle band (two halo rows), a top band and a bottom band"""
    from malstroem_amd.distributed import HipBand
    dem = fbm(150, 70, beta=2.0, seed=6)
    bands = [HipBand(150, 70, r0, 50, device=0, rank=k, size=3) for k, r0 in enumerate((0, 50, 100))]
    try:
        for b, r0 in zip(bands, (0, 50, 100)):
            b.upload("dem", dem[r0:r0 + 50])
        top, mid, bot = bands
        first, last = mid.get_edge_rows("dem")
        assert np.array_equal(first, dem[50]) and np.array_equal(last, dem[99])
        assert mid.get_edge_rows("dem", first=False)[0] is None and np.array_equal(top.get_edge_rows("dem", first=False)[1], dem[49])
        assert mid.set_halo_rows("dem", dem[49], dem[100]) == (True, True)
        assert np.array_equal(mid.get_edge_row("dem", 2), dem[49]) and np.array_equal(mid.get_edge_row("dem", 3), dem[100])
        assert mid.set_halo_rows("dem", dem[49], dem[100]) == (False, False)          # the same bytes again
        other = dem[100].copy()
        other[3] += 1.0
        assert mid.set_halo_rows("dem", dem[49], other) == (False, True)
        assert mid.set_halo_rows("dem", None, None) == (False, False)
        assert top.set_halo_rows("dem", None, dem[50]) == (False, True) and np.array_equal(top.get_edge_row("dem", 3), dem[50])
        assert bot.set_halo_rows("dem", dem[99], None) == (True, False) and np.array_equal(bot.get_edge_row("dem", 2), dem[99])
        with pytest.raises(ValueError):
            top.set_halo_rows("dem", dem[0], None)                                     # no halo row above the first band
    finally:
        for b in bands:
            b.close()
