"""GPU: edge cases and randomised inputs beyond the D8-of-a-no-flats-surface pipeline, always against the oracle
(or scipy for the labelling): flow cycles, interior sinks, inward pointing edges, tiny / ragged rasters, masks with
every kind of adjacency, explicit nlabels, error behaviour."""
import math

import numpy as np
import pytest
import scipy.ndimage

import oracle
from _cases import fbm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def alg():
    import malstroem_amd.algorithms as a
    assert a.hip.available
    return a


def random_acyclic_flow(rng, h, w, edges, sink_frac=0.03):
    """Random D8 codes that always descend a random strictly ordered potential => acyclic; `edges`: outward / nodir / free."""
    pot = rng.permutation(h * w).reshape(h, w).astype(np.int64)
    fd = np.full((h, w), 8, np.uint8)
    DR = [-1, -1, 0, 1, 1, 1, 0, -1]
    DC = [0, 1, 1, 1, 0, -1, -1, -1]
    for r in range(h):
        for c in range(w):
            cand = [k for k in range(8) if 0 <= r + DR[k] < h and 0 <= c + DC[k] < w and pot[r + DR[k], c + DC[k]] < pot[r, c]]
            if cand and rng.random() > sink_frac:
                fd[r, c] = cand[rng.integers(len(cand))]
    if edges == "outward":
        from malstroem_amd.algorithms.flow import set_edges_flow_outward
        set_edges_flow_outward(fd)
    elif edges == "nodir":
        fd[0, :] = fd[-1, :] = 8
        fd[:, 0] = fd[:, -1] = 8
    return fd


@pytest.mark.parametrize("edges", ["outward", "nodir", "free"])
@pytest.mark.parametrize("seed", range(6))
def test_watersheds_on_random_acyclic_flow_with_sinks(alg, edges, seed):
    rng = np.random.default_rng(100 + seed)
    h, w = int(rng.integers(3, 40)), int(rng.integers(3, 40))
    fd = random_acyclic_flow(rng, h, w, edges)
    lab = np.where(rng.random((h, w)) < 0.15, rng.integers(1, 9, (h, w)), 0).astype(np.int32)
    want = lab.copy()
    oracle.watersheds_from_labels(fd, want, 0)
    got = lab.copy()
    alg.flow.watersheds_from_labels(fd, got, 0)
    assert np.array_equal(got, want)
    # a non-zero `unassigned` marker
    lab2 = np.where(lab == 0, -7, lab).astype(np.int32)
    want2, got2 = lab2.copy(), lab2.copy()
    oracle.watersheds_from_labels(fd, want2, -7)
    alg.flow.watersheds_from_labels(fd, got2, -7)
    assert np.array_equal(got2, want2)


@pytest.mark.parametrize("seed", range(6))
def test_accumulation_on_random_flow_incl_cycles(alg, seed):
    rng = np.random.default_rng(200 + seed)
    h, w = int(rng.integers(2, 70)), int(rng.integers(2, 140))
    fd = rng.integers(0, 9, (h, w)).astype(np.uint8)     # arbitrary codes: cycles, sinks, off-raster flow
    assert np.array_equal(alg.flow.accumulated_flow(fd), oracle.accumulated_flow(fd))
    fd2 = random_acyclic_flow(rng, h, w, "free", sink_frac=0.0)
    acc = alg.flow.accumulated_flow(fd2)
    assert np.array_equal(acc, oracle.accumulated_flow(fd2)) and acc.min() >= 1


def test_watersheds_terminate_on_a_cycle_through_edge_cells(alg):
    fd = np.full((4, 5), 8, np.uint8)
    fd[0, 1], fd[0, 2] = 2, 6          # two edge cells pointing at each other: the reference never returns
    fd[1, 1] = 0                       # drains into the cycle
    lab = np.zeros((4, 5), np.int32)
    lab[3, 3] = 5
    alg.flow.watersheds_from_labels(fd, lab, 0)
    assert lab[0, 1] == 0 and lab[0, 2] == 0 and lab[1, 1] == 0 and lab[3, 3] == 5


@pytest.mark.parametrize("shape", [(1, 1), (1, 9), (7, 1), (2, 2), (3, 3), (3, 130), (130, 3), (65, 63), (64, 64), (63, 127)])
def test_tiny_and_ragged_rasters(alg, shape):
    rng = np.random.default_rng(sum(shape))
    dem = (rng.random(shape) * 10).astype(np.float32)
    filled = alg.fill.fill_terrain(dem)
    assert np.array_equal(filled, oracle.fill_terrain(dem))
    short, diag = alg.fill.minimum_safe_short_and_diag(dem)
    assert (short, diag) == oracle.minimum_safe_short_and_diag(dem)
    fnf = alg.fill.fill_terrain_no_flats(dem, short, diag)
    assert np.array_equal(fnf, oracle.fill_terrain_no_flats(dem, short, diag))
    for outward in (True, False):
        assert np.array_equal(alg.flow.terrain_flowdirection(fnf, outward), oracle.terrain_flowdirection(fnf, outward))
    fd = alg.flow.terrain_flowdirection(fnf)
    assert np.array_equal(alg.flow.accumulated_flow(fd), oracle.accumulated_flow(fd))
    lab, n = alg.label.connected_components(alg.fill.bluespot_depths(filled, dem))
    olab, on = oracle.connected_components(oracle.depths(filled, dem))
    assert n == on and np.array_equal(lab, olab)
    assert np.array_equal(alg.label.label_count(lab), np.bincount(lab.ravel()))


def test_no_flats_defaults_equal_plain_fill_in_float64(alg):
    rng = np.random.default_rng(5)
    dem = (rng.random((50, 60)) * 5).astype(np.float32)
    assert np.array_equal(alg.fill.fill_terrain_no_flats(dem), oracle.fill_terrain_no_flats(dem, 0.0, 0.0))
    assert np.array_equal(alg.fill.fill_terrain_no_flats(dem), alg.fill.fill_terrain(dem).astype(np.float64))


def test_fill_with_nan_cells_matches_the_reference_comparisons(alg):
    rng = np.random.default_rng(6)
    dem = (rng.random((40, 45)) * 5).astype(np.float32)
    dem[10, 10] = dem[20:23, 30] = np.nan          # interior NaN: never updated, never wins a min (fill.py:37, _fill.pyx:48)
    got, want = alg.fill.fill_terrain(dem), oracle.fill_terrain(dem)
    assert np.array_equal(got, want, equal_nan=True)


def test_fill_with_nan_on_the_raster_border(alg):
    """Border cells are copied from the DEM (fill.py:102-109): a NaN border cell stays NaN and never wins a min
    (`a <= NaN` is false, _fill.pyx:22), the interior behind it fills as if it were a wall."""
    rng = np.random.default_rng(16)
    for shape in ((40, 45), (70, 130)):
        dem = (rng.random(shape) * 5).astype(np.float32)
        dem[0, 3] = dem[0, 0] = dem[-1, 7:12] = dem[5:9, 0] = dem[17, -1] = dem[-1, -1] = np.nan
        dem[1, 3] = dem[6, 1] = 0.0                      # pits right behind NaN border cells
        got, want = alg.fill.fill_terrain(dem), oracle.fill_terrain(dem)
        assert np.isnan(got[0, 3]) and np.isnan(got[-1, -1]) and np.isnan(got[6, 0])
        assert np.array_equal(got, want, equal_nan=True)
        # no-flats: the reference is not defined on NaN -- its nested two-argument minima are order dependent there and
        # its two own variants disagree with each other (Cython `a <= b ? a : b`, _fill.pyx:25: a NaN in the LAST diagonal
        # silences all four diagonals; Python's builtin min, fill.py:87-88: a NaN in the FIRST one does); malstroem
        # declares nodata unsupported (fill.py:118, the CLI substitutes -999, io.py:69-71).  The kernel's rule is the plain
        # fill's: a NaN cell never wins a minimum (== +inf) and a NaN border cell stays NaN.
        short, diag = oracle.minimum_safe_short_and_diag(np.nan_to_num(dem))
        got = alg.fill.fill_terrain_no_flats(dem, short, diag)
        want = oracle.fill_terrain_no_flats(np.where(np.isnan(dem), np.float32(np.inf), dem), short, diag)
        border = np.ones(dem.shape, bool)
        border[1:-1, 1:-1] = False
        want[np.isnan(dem) & border] = np.nan            # border cells are copies of the DEM
        assert np.isnan(got[0, 3]) and np.isnan(got[-1, -1]) and np.isnan(got[6, 0])
        assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("seed", range(5))
def test_connected_components_random_masks_equal_scipy(alg, seed):
    rng = np.random.default_rng(300 + seed)
    h, w = int(rng.integers(1, 200)), int(rng.integers(1, 300))
    for density in (0.05, 0.4, 0.6, 0.95):
        mask = rng.random((h, w)) < density
        want, n = scipy.ndimage.label(mask, structure=np.ones((3, 3)))
        for data in (mask, mask.astype(np.uint8) * 3, np.where(mask, rng.standard_normal((h, w)), 0).astype(np.float32)):
            got, gn = alg.label.connected_components(data)
            assert got.dtype == np.int32 and gn == n and np.array_equal(got, want)
    full = np.ones((h, w), np.float32)
    full[0, 0] = np.nan                                   # NaN != 0: foreground like in scipy
    got, gn = alg.label.connected_components(full)
    assert gn == 1 and got.min() == 1
    got, gn = alg.label.connected_components(np.zeros((h, w), np.float32))
    assert gn == 0 and not got.any()


def test_label_reductions_with_explicit_nlabels_ties_and_nan(alg):
    rng = np.random.default_rng(7)
    lab = rng.integers(0, 6, (70, 90)).astype(np.int32)
    data32 = np.round(rng.standard_normal((70, 90)), 1).astype(np.float32)     # many exact ties
    data64 = data32.astype(np.float64)
    data64[lab == 4] = 1.5                                                     # constant label: first raster cell wins
    st, ost = alg.label.label_stats(data32, lab, 9), oracle.label_stats(data32, lab, 9)   # nlabels > max(label)
    assert len(st) == 10
    for f in ("min", "max", "count"):
        assert np.array_equal(st[f], ost[f])
    assert np.allclose(st["sum"], ost["sum"], rtol=1e-12, atol=0)
    for fn, ofn in ((alg.label.label_min_index, oracle.label_min_index), (alg.label.label_max_index, oracle.label_max_index)):
        a, b = fn(data64, lab, 9), ofn(data64, lab, 9)
        for f in ("value", "row", "col"):
            assert np.array_equal(a[f], b[f]), f
        assert a["row"][7] == -1 and a["col"][9] == -1                         # empty labels keep (inf, -1, -1)
    d = data64.copy()
    d[lab == 2] = np.nan                                                       # all-NaN label: never selected
    a, b = alg.label.label_min_index(d, lab), oracle.label_min_index(d, lab)
    assert a["row"][2] == -1 and np.array_equal(a["row"], b["row"]) and np.array_equal(a["col"], b["col"])
    with pytest.raises(ValueError):
        alg.label.label_stats(data32, lab, 3)                                  # label outside [0, nlabels]


def test_keep_labels_mutates_and_masks_like_the_reference(alg):
    lab = np.array([[0, 1, 1], [2, 0, 3]], np.int32)
    keep = [True, True, False, True]
    mask = alg.label.keep_labels(lab, keep)
    assert keep[0] is False                                                    # label.py:91 side effect
    assert mask.dtype == bool and mask.tolist() == [[False, True, True], [False, False, True]]


def test_fills_on_shapes_around_tile_and_macro_tile_multiples():
    """62-cell tiles, 124-cell macro tiles: shapes right at / next to the multiples (ragged macro tiles with 1, 2 or 3 real
    tiles, windows that end on the raster border, one-tile rasters), both fills against the oracle."""
    import oracle
    import malstroem_amd.algorithms as alg
    rng = np.random.default_rng(77)
    dims = [3, 4, 63, 64, 65, 66, 125, 126, 127, 128, 187, 188, 189, 250, 251]
    shapes = [(h, w) for h in dims for w in (64, 126, 127, 189)] + [(251, 250), (126, 3), (4, 190)]
    for h, w in shapes:
        # rough surface with a few deep pits and plateaus: many small lakes and some that span tiles
        dem = rng.normal(size=(h, w)).cumsum(0).cumsum(1).astype(np.float32)
        dem -= dem.min()
        dem *= 50.0 / max(float(dem.max()), 1.0)
        dem = np.round(dem, 1).astype(np.float32)          # plateaus (ties)
        if h > 8 and w > 8:
            dem[h // 2 - 2:h // 2 + 2, 1:w - 1] = -5.0       # a trench across all tile columns
        filled = alg.fill.fill_terrain(dem)
        assert np.array_equal(filled, oracle.fill_terrain(dem)), (h, w)
        short, diag = alg.fill.minimum_safe_short_and_diag(dem)
        got = alg.fill.fill_terrain_no_flats(dem, short, diag)
        assert np.array_equal(got, oracle.fill_terrain_no_flats(dem, short, diag)), (h, w)


def test_label_stats_on_non_float32_rasters(alg):
    """The reference's generic path (label.py:43-75) takes any raster; float64 / integer data run the float64 kernel."""
    rng = np.random.default_rng(21)
    lab = rng.integers(0, 40, size=(90, 131)).astype(np.int32)
    lab[10:40, 20:90] = 0
    for data in (rng.standard_normal(lab.shape), rng.integers(-50, 50, size=lab.shape), rng.integers(0, 255, size=lab.shape).astype(np.uint8),
                 (rng.random(lab.shape) * 1e-3).astype(np.float16)):
        got = alg.label.label_stats(data, lab)
        d64 = data.astype(np.float64)
        for l in range(40):
            sel = d64[lab == l]
            assert got["count"][l] == sel.size and got["min"][l] == sel.min() and got["max"][l] == sel.max()
            assert abs(got["sum"][l] - math.fsum(sel)) <= 1e-12 * max(1.0, np.abs(sel).sum())
    data = rng.standard_normal(lab.shape)
    data[3, 3] = np.nan
    got, want = alg.label.label_stats(data, lab, 45), oracle.label_stats(data.astype(np.float32), lab, 45)
    assert len(got) == 46 and np.isnan(got["sum"][lab[3, 3]]) and got["count"][45] == 0 and got["min"][45] == np.inf
    assert np.array_equal(got["count"], want["count"])


def test_priority_flood_and_its_fallback_give_the_same_fill(alg):
    """fill_terrain runs the tiled priority-flood (pflood.hip); a tile that exceeds one of its capacities (here: a pit in
    every other cell -> more adjacent basin pairs than the LDS hash holds) sends the whole raster through the iterative
    tile schedule instead.  Both paths and the oracle agree bit for bit; the context reports which one ran."""
    from malstroem_amd.pipeline import HydroPipeline
    rng = np.random.default_rng(33)
    smooth = fbm(300, 260, beta=2.0, seed=12)
    pits = (1.0 + rng.random((300, 260))).astype(np.float32)
    pits[::2, ::2] = (rng.random((150, 130)) * 0.5).astype(np.float32)
    plateau = np.round(fbm(200, 330, beta=2.5, seed=13) / 4).astype(np.float32)     # integer steps: large plateaus
    for dem, algorithm in ((smooth, 1), (pits, 0), (plateau, 1)):
        with HydroPipeline(dem.shape) as pipe:
            pipe.upload("dem", dem)
            pipe.run("fill")
            pipe.sync()
            got, dep = pipe.download("filled"), pipe.download("depths")
            assert pipe.get_int("fill_algorithm") == algorithm
        want = oracle.fill_terrain(dem)
        assert np.array_equal(got, want)
        assert np.array_equal(dep, oracle.depths(want, dem))


def test_the_flood_proof_is_live(alg, monkeypatch):
    """Every plain fill by the tiled priority-flood is proven at run time (check.hip: the reference's update evaluated at every
    cell of the result).  MHIP_PF_CORRUPT raises one interior cell after the flood: the check has to see it, the iterative
    schedule repairs the surface from there (fill_algorithm 4 = flood + repair) and the oracle's bits come out."""
    from malstroem_amd.pipeline import HydroPipeline
    dem = fbm(420, 380, beta=2.0, seed=21)
    want = oracle.fill_terrain(dem)
    for corrupt, algorithm in ((False, 1), (True, 4)):
        if corrupt:
            monkeypatch.setenv("MHIP_PF_CORRUPT", "1")
        with HydroPipeline(dem.shape) as pipe:
            pipe.upload("dem", dem)
            pipe.run("fill")
            pipe.sync()
            assert pipe.get_int("fill_algorithm") == algorithm
            assert np.array_equal(pipe.download("filled"), want)
            assert np.array_equal(pipe.download("depths"), oracle.depths(want, dem))
    assert np.array_equal(alg.fill.fill_terrain(dem), want)      # (stage function, hook still set)


@pytest.mark.parametrize("shape,beta,seed", [((300, 256), 2.0, 1), ((700, 512), 2.5, 2), ((130, 1024), 2.0, 3), ((64, 768), 1.5, 4), ((2, 256), 2.0, 5),
                                             ((3, 512), 2.0, 6), ((1000, 1280), 3.0, 7)])
def test_fused_apply_and_proof_on_widths_that_are_multiples_of_256(alg, shape, beta, seed):
    """rasters whose width is a multiple of 256 take pf_apply_check_kernel (K4 and the run-time proof in one streaming pass)
    instead of pf_apply_kernel + fill_check_kernel: same filled surface and depths, incl. NaN cells, -0.0 and tiny heights"""
    from malstroem_amd.pipeline import HydroPipeline
    dem = fbm(shape[0], shape[1], beta=beta, seed=seed)
    if shape[0] > 60:
        dem[shape[0] // 2, 17] = np.nan
        dem[shape[0] // 3, 250:262] = -0.0
    want = oracle.fill_terrain(dem)
    with HydroPipeline(dem.shape) as pipe:
        pipe.upload("dem", dem)
        pipe.run("fill")
        pipe.sync()
        got, dep = pipe.download("filled"), pipe.download("depths")
        assert pipe.get_int("fill_algorithm") in (0, 1)       # (4 = the proof failed and the surface was repaired: never here)
    assert np.array_equal(got, want, equal_nan=True)
    assert np.array_equal(dep, oracle.depths(want, dem), equal_nan=True)


def test_pour_points_out_of_the_accumulation_pass_and_the_cases_that_take_the_general_pass():
    """ACCUM + WATERSHED + POURPOINTS in one request: the watersheds' tile pass lists the candidate cells, the accumulation's final
    pass turns them into keys (common.hpp: PourLink).  Labels that are no connected components (the candidate test compares
    labels), a flow cycle (unresolved cells: the general pass), more than 512 labelled candidates in a tile (overflow: the
    general pass) -- the records are label_max_index(accumulated_flow) of the oracle every time, record 0 included."""
    from malstroem_amd.pipeline import HydroPipeline
    from malstroem_amd.algorithms.flow import set_edges_flow_outward
    from _cases import zigzag_flowdir

    def run(fd, labels, dem=None):
        with HydroPipeline(labels.shape) as p:
            if dem is not None:      # directions the library computed itself: no flow cycle, the watersheds' fast path
                p.upload("dem", dem)
                p.run("fill", "noflat", "flowdir")
            else:                    # of unknown origin: the general path (a labelled cell on a cycle labels nothing)
                p.upload("flowdir", fd)
            p.upload("labels", labels)
            for _ in range(2):
                p.run("accum", "watershed", "pourpoints")
            p.sync()
            return p.download("accum"), p.download("watersheds"), p.pourpoints(), p.get_int("pour_algorithm")

    def check(fd, labels, from_keys, dem=None):
        n = int(labels.max())
        acc, ws, pour, alg = run(fd, labels, dem)
        assert alg == from_keys
        oacc = oracle.accumulated_flow(fd)
        ows = labels.copy()
        oracle.watersheds_from_labels(fd, ows)
        assert np.array_equal(acc, oacc) and np.array_equal(ws, ows)
        opour = oracle.label_max_index(oacc, labels, n)
        for f in opour.dtype.names:
            assert np.array_equal(pour[f], opour[f]), f

    rng = np.random.default_rng(5)
    h, w = 300, 520
    dem = fbm(h, w, seed=11)
    s, d = oracle.minimum_safe_short_and_diag(dem)
    fd = oracle.terrain_flowdirection(oracle.fill_terrain_no_flats(dem, s, d))
    # (a) rectangles of labels: neighbours with different labels, one label in several places, unlabelled gaps
    labels = np.zeros((h, w), np.int32)
    for k in range(400):
        r, c = int(rng.integers(0, h - 8)), int(rng.integers(0, w - 8))
        labels[r:r + int(rng.integers(1, 9)), c:c + int(rng.integers(1, 9))] = int(rng.integers(1, 120))
    check(fd, labels, 1, dem)
    check(fd, labels, 0)
    # (b) every cell flows to the right, two cells in the middle flow into each other: everything upstream of them is unresolved
    fdz = zigzag_flowdir(h, w, 3)
    set_edges_flow_outward(fdz)
    fdz[150, 200], fdz[150, 201] = 2, 6
    cyc = labels.copy()
    cyc[150, 200:202] = 121         # a label that is the cycle alone: no candidate cell, its record is (0, first cell)
    check(fdz, cyc, 0)
    # (c) a label of its own for every cell of a block: every one of them is a candidate
    many = np.zeros((h, w), np.int32)
    many[64:128, 64:128] = 1 + np.arange(64 * 64, dtype=np.int32).reshape(64, 64)
    check(fd, many, 0, dem)
