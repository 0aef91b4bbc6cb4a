"""GPU parity tests: every HIP stage (through the C-ABI / malstroem_amd.algorithms) against
  * the reference's golden rasters and the pure-Python reference goldens (bit-exact), and
  * the CPU oracle on seeded synthetic inputs at sizes the oracle finishes in seconds.
Tolerances: bit-exact everywhere (uint8 flow directions, int32 labels / watersheds, integer-valued float64
accumulation, float32 filled surface, float64 no-flats surface); label_stats `sum` is compared exactly too
and, where the reference's own sequential sum is inexact, within 1e-12 relative.
"""
import numpy as np
import pytest

import oracle
from _cases import (PYREF_CASES, assemble_reference_pourpoints, assert_label_sums, d8_mul_vs_div_case, fbm, fixtures,
                    pyref, reference_vectors)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def alg():
    import malstroem_amd.algorithms as a
    assert a.hip.available, "HIP library / device missing: the product has no CPU fallback"
    return a


@pytest.fixture(scope="module")
def fx():
    return fixtures()


def test_native_library_is_loaded(alg):
    from malstroem_amd import _lib
    assert _lib.load().mhip_device_count() >= 1
    assert b"gfx950" in _lib.load().mhip_version()


# ---- reference fixtures (reference tests/test_raster_{fill,flowdir,label}.py) -------------------------

def test_fill_reference_golden(alg, fx):
    filled = alg.fill.fill_terrain(fx["dtm"])
    assert filled.dtype == np.float32 and np.array_equal(filled, fx["filled"])
    assert filled.max() == fx["dtm"].max()


def test_noflats_reference_golden(alg, fx):
    short, diag = alg.fill.minimum_safe_short_and_diag(fx["dtm"])
    assert (short, diag) == (7.275957614183426e-12, 1.0289757937229989e-11)
    fnf = alg.fill.fill_terrain_no_flats(fx["dtm"], short, diag)
    assert fnf.dtype == np.float64 and np.array_equal(fnf, fx["filled_no_flats"])
    assert fnf.max() <= fx["dtm"].max() + sum(fx["dtm"].shape) * diag


def test_depths_reference_golden(alg, fx):
    assert np.array_equal(alg.fill.bluespot_depths(fx["filled"], fx["dtm"]), fx["depths"])


def test_flowdir_reference_golden(alg, fx):
    fd = alg.flow.terrain_flowdirection(fx["filled_no_flats"])
    assert fd.dtype == np.uint8 and np.array_equal(fd, fx["flowdir_noflats"])
    with pytest.raises(ValueError):
        alg.flow.terrain_flowdirection(fx["filled"])   # float32 is rejected like _flow.pyx:99


def test_trace_and_upstream_known_answers(alg, fx):
    # reference tests/test_raster_flowdir.py:28-46
    fd = alg.flow.terrain_flowdirection(fx["filled_no_flats"])
    trace = list(alg.flow.trace_downstream(fd, (100, 100)))
    assert len(trace) == 100 and trace[-1] == (187, 83)
    seen, stack = set(), [(186, 82)]
    while stack:
        c = stack.pop()
        seen.add(c)
        stack.extend(alg.flow.upstream_cells(fd, c))
    assert len(seen) == 5267


def test_accum_reference_known_answers(alg, fx):
    acc = alg.flow.accumulated_flow(fx["flowdir_noflats"])
    assert acc.dtype == np.float64
    assert acc.min() >= 1 and acc.max() == 11158 and acc.sum() == 3578615
    assert np.array_equal(acc, oracle.accumulated_flow(fx["flowdir_noflats"]))


def test_watersheds_reference_golden(alg, fx):
    for dtype in (np.int32, np.int64, np.uint32):   # reference tests/test_raster_flowdir.py:73-140
        ws = fx["labelled"].astype(dtype)
        alg.flow.watersheds_from_labels(fx["flowdir_noflats"], ws, unassigned=0)
        assert ws.dtype == dtype and np.array_equal(ws, fx["wsheds"])
    assert ws.sum() == 2337891


def test_ccl_reference_known_answers(alg, fx):
    lab, n = alg.label.connected_components(fx["filled_no_flats"] - fx["filled"])
    assert lab.dtype == np.int32 and n == 525
    assert (lab == 0).sum() == 40029 and lab.sum() == 1561377


def test_label_stats_reference(alg, fx):
    st = alg.label.label_stats(fx["depths"], fx["labelled"])
    ref = oracle.label_stats(fx["depths"], fx["labelled"])
    assert len(st) == fx["labelled"].max() + 1 and st["count"].sum() == fx["labelled"].size
    for f in ("min", "max", "sum", "count"):
        assert np.array_equal(st[f], ref[f]), f


def test_label_index_reference(alg, fx):
    acc = oracle.accumulated_flow(fx["flowdir_noflats"])
    mi = alg.label.label_min_index(fx["filled_no_flats"], fx["labelled"])
    ma = alg.label.label_max_index(acc, fx["labelled"], int(fx["labelled"].max()))
    rmi = oracle.label_min_index(fx["filled_no_flats"], fx["labelled"])
    rma = oracle.label_max_index(acc, fx["labelled"])
    for f in ("value", "row", "col"):
        assert np.array_equal(mi[f], rmi[f]) and np.array_equal(ma[f], rma[f]), f


def test_flowdir_follows_the_cython_variant_where_the_two_differ(alg):
    """_flow.pyx:93-94,140 multiplies the diagonal drop by 1/2**0.5, flow.py:79 divides by sqrt(2): on this surface the
    two give different codes at the centre cell; the kernel must give the Cython one."""
    z, code_cython, code_python = d8_mul_vs_div_case()
    assert oracle.terrain_flowdirection(z, variant="python")[1, 1] == code_python != code_cython
    for pad in (0, 5, 70):     # the cell inside a 3x3 raster, and inside rasters wider than one lane strip
        zz = np.zeros((3 + 2 * pad, 3 + 2 * pad))
        zz[pad:pad + 3, pad:pad + 3] = z
        fd = alg.flow.terrain_flowdirection(zz)
        assert fd[pad + 1, pad + 1] == code_cython
        assert np.array_equal(fd, oracle.terrain_flowdirection(zz, variant="cython"))


def test_pourpoints_reproduce_the_reference_vector_fixture(alg, fx):
    """reference tests/data/pourpoints.json: 105 records = label_min_index(no-flats surface) + label_stats(depths) +
    label_count(watersheds) (bluespots.py:49-88, 195-206, the branch WITHOUT accumulated flow).  Three ways: the stage
    functions, the device pipeline's POURPOINTS stage on its arg-min branch, and BluespotTool given input_dem only."""
    from malstroem_amd.bluespots import BluespotTool, assemble_pourpoints
    from malstroem_amd.pipeline import HydroPipeline
    ref = reference_vectors()["pourpoints"]
    gt = [float(v) for v in fx["geotransform"]]
    lab = fx["labelled"]

    def check(features):
        assert len(features) == len(ref) == 105
        for f, r in zip(features, ref):
            p = f["properties"] if "properties" in f else f
            for k in ("bspot_id", "cell_row", "cell_col", "bspot_dmax", "bspot_area", "bspot_vol", "wshed_area"):
                assert p[k] == r["properties"][k], (p["bspot_id"], k)
            if "geometry" in f:
                assert np.allclose(f["geometry"]["coordinates"], r["coordinates"], rtol=0, atol=1e-6)

    # (1) stage functions
    pp = alg.label.label_min_index(fx["filled_no_flats"], lab)
    st = alg.label.label_stats(fx["depths"], lab)
    ws = lab.copy()
    alg.flow.watersheds_from_labels(fx["flowdir_noflats"], ws, 0)
    check(assemble_pourpoints(gt, pp, st, alg.label.label_count(ws)))
    check(assemble_reference_pourpoints(abs(gt[1]) * abs(gt[5]), pp, st, alg.label.label_count(ws)))

    # (2) device pipeline: labels uploaded, no accumulated flow -> POURPOINTS takes arg-min of the no-flats surface
    with HydroPipeline(lab.shape) as pipe:
        pipe.upload("dem", fx["dtm"])
        pipe.upload("depths", fx["depths"])
        pipe.upload("flowdir", fx["flowdir_noflats"])
        pipe.run("noflat")
        pipe.upload("labels", lab)
        pipe.run("watershed", "pourpoints")
        pix, counts = pipe.pourpoints(), pipe.watershed_counts()
        assert np.array_equal(pipe.download("watersheds"), fx["wsheds"])
    check(assemble_pourpoints(gt, pix, st, counts))

    # (3) BluespotTool, input_dem given, no input_accum; the filter that made labelled.tif is not recorded, so it is
    #     rebuilt from the fixture itself: keep the raw bluespots that survive in labelled.tif
    raw, nraw = oracle.connected_components(fx["depths"])
    kept = np.unique(raw[lab > 0])
    rst = oracle.label_stats(fx["depths"], raw)
    keys = {(rst["min"][l], rst["max"][l], rst["sum"][l], int(rst["count"][l])) for l in kept}

    class Reader(object):
        def __init__(self, a):
            self.a, self.transform = a, gt

        def read(self):
            return self.a

    class Writer(object):
        def write(self, a):
            self.a = a

        def write_geojson_features(self, fc):
            self.fc = fc

    labeled, pour, wsheds = Writer(), Writer(), Writer()
    BluespotTool(input_depths=Reader(fx["depths"]), input_flowdir=Reader(fx["flowdir_noflats"]),
                 input_bluespot_filter_function=lambda s: (s["min"], s["max"], s["sum"], int(s["count"])) in keys,
                 input_dem=Reader(fx["dtm"]), output_labeled_raster=labeled, output_pourpoints=pour,
                 output_watersheds_raster=wsheds).process()
    assert np.array_equal(labeled.a, lab) and np.array_equal(wsheds.a, fx["wsheds"])
    check(pour.fc["features"])


# ---- pure-Python reference goldens ---------------------------------------------------------------------

@pytest.mark.parametrize("name", PYREF_CASES)
def test_chain_equals_python_reference(alg, name):
    g = pyref(name)
    dem = g["dem"]
    filled = alg.fill.fill_terrain(dem)
    assert np.array_equal(filled, g["filled"])
    assert np.array_equal(alg.fill.bluespot_depths(filled, dem), g["depths"])
    short, diag = alg.fill.minimum_safe_short_and_diag(dem)
    assert [short, diag] == g["short_diag"].tolist()
    fnf = alg.fill.fill_terrain_no_flats(dem, short, diag)
    assert np.array_equal(fnf, g["filled_no_flats"])
    fd = alg.flow.terrain_flowdirection(fnf)
    assert np.array_equal(fd, g["flowdir"])   # multiply (cython) == divide (python) on these inputs, see oracle test
    assert np.array_equal(alg.flow.terrain_flowdirection(fnf, edges_flow_outward=False), g["flowdir_edges_nodir"])
    assert np.array_equal(alg.flow.terrain_flowdirection(filled.astype(np.float64)), g["flowdir_of_filled"])
    assert np.array_equal(alg.flow.accumulated_flow(fd), g["accum"])
    raw, n = alg.label.connected_components(g["depths"])
    assert n == int(g["raw_nlabels"]) and np.array_equal(raw, g["raw_labels"])
    st = alg.label.label_stats(g["depths"], raw)
    for f in ("min", "max", "sum", "count"):
        assert np.array_equal(st[f], g["raw_stats"][f]), f
    mask = alg.label.keep_labels(raw, list(g["keepers"]))
    assert mask.dtype == bool and np.array_equal(mask, g["keep_mask"])
    lab, nl = alg.label.connected_components(mask)
    assert nl == int(g["nlabels"]) and np.array_equal(lab, g["labeled"])
    ws = lab.copy()
    alg.flow.watersheds_from_labels(fd, ws, 0)
    assert np.array_equal(ws, g["watersheds"])
    assert np.array_equal(alg.label.label_count(ws), g["watershed_counts"])
    mi = alg.label.label_min_index(fnf, lab, nl)
    ma = alg.label.label_max_index(g["accum"], lab, nl)
    for f in ("value", "row", "col"):
        assert np.array_equal(mi[f], g["min_index"][f]), f
        assert np.array_equal(ma[f], g["max_index"][f]), f


# ---- seeded synthetic inputs vs the oracle -----------------------------------------------------------------

SYNTH = [("fbm512_b2", 512, 512, 2.0, 42), ("fbm700x450_b3", 700, 450, 3.0, 7), ("fbm1024_b2", 1024, 1024, 2.0, 1)]


@pytest.mark.parametrize("name,h,w,beta,seed", SYNTH)
def test_chain_equals_oracle(alg, name, h, w, beta, seed):
    dem = fbm(h, w, beta=beta, seed=seed)
    filled = alg.fill.fill_terrain(dem)
    assert np.array_equal(filled, oracle.fill_terrain(dem))
    short, diag = alg.fill.minimum_safe_short_and_diag(dem)
    assert (short, diag) == oracle.minimum_safe_short_and_diag(dem)
    fnf = alg.fill.fill_terrain_no_flats(dem, short, diag)
    assert np.array_equal(fnf, oracle.fill_terrain_no_flats(dem, short, diag))
    fd = alg.flow.terrain_flowdirection(fnf)
    assert np.array_equal(fd, oracle.terrain_flowdirection(fnf))
    acc = alg.flow.accumulated_flow(fd)
    assert np.array_equal(acc, oracle.accumulated_flow(fd))
    depths = alg.fill.bluespot_depths(filled, dem)
    assert np.array_equal(depths, oracle.depths(filled, dem))
    raw, n = alg.label.connected_components(depths)
    oraw, on = oracle.connected_components(depths)
    assert n == on and np.array_equal(raw, oraw)
    st, ost = alg.label.label_stats(depths, raw), oracle.label_stats(depths, raw)
    for f in ("min", "max", "count"):
        assert np.array_equal(st[f], ost[f]), f
    assert_label_sums(st["sum"], ost["sum"], depths, raw)   # exact wherever the reference's own sum is (fsum guard)
    keep = (ost["count"] >= 5) & (ost["max"] > 0.05)
    lab, nl = alg.label.connected_components(alg.label.keep_labels(raw, list(keep)))
    olab, onl = oracle.connected_components(oracle.keep_labels(oraw, list(keep)))
    assert nl == onl and np.array_equal(lab, olab)
    ws, ows = lab.copy(), olab.copy()
    alg.flow.watersheds_from_labels(fd, ws, 0)
    oracle.watersheds_from_labels(fd, ows, 0)
    assert np.array_equal(ws, ows)
    assert np.array_equal(alg.label.label_count(ws), oracle.label_count(ows))
    for a, b in ((alg.label.label_min_index(fnf, lab, nl), oracle.label_min_index(fnf, lab, nl)),
                 (alg.label.label_max_index(acc, lab, nl), oracle.label_max_index(acc, lab, nl))):
        for f in ("value", "row", "col"):
            assert np.array_equal(a[f], b[f]), f


# ---- drop-in boundary: the hip switch rebinds a target package like speedups.enable() does ---------------------

def test_enable_patches_and_disable_restores_a_target_package(alg):
    import types
    sentinel = object()
    target = types.SimpleNamespace(
        fill=types.SimpleNamespace(fill_terrain=sentinel, fill_terrain_no_flats=sentinel, minimum_safe_short_and_diag=sentinel),
        flow=types.SimpleNamespace(terrain_flowdirection=sentinel, _terrain_flow=sentinel, accumulated_flow=sentinel,
                                   watersheds_from_labels=sentinel),
        label=types.SimpleNamespace(connected_components=sentinel, label_stats=sentinel, label_min_index=sentinel,
                                    label_max_index=sentinel, keep_labels=sentinel, label_count=sentinel))
    alg.hip.disable()
    alg.hip.enable(target=target)
    try:
        assert alg.hip.enabled and alg.speedups.enabled
        assert target.fill.fill_terrain is alg.fill.fill_terrain
        assert target.flow.accumulated_flow is alg.flow.accumulated_flow
        assert target.label.connected_components is alg.label.connected_components
        alg.hip.enable(target=target)  # idempotent like speedups.enable (speedups/__init__.py:44-45)
    finally:
        alg.hip.disable()
    assert target.fill.fill_terrain is sentinel and target.label.label_count is sentinel
    assert not alg.hip.enabled
    alg.hip.enable()


def test_tools_run_on_the_device_pipeline(alg, fx):
    """DemTool + BluespotTool (reference tests/test_raster_dem.py:7-22, test_raster_bluespot.py:18-69) with in-memory
    reader/writer stubs like the reference's NumpyRasterReader (test_raster_bluespot.py:9-15)."""
    from malstroem_amd.bluespots import BluespotTool
    from malstroem_amd.dem import DemTool

    class Reader(object):
        def __init__(self, a):
            self.a, self.transform = a, [float(v) for v in fx["geotransform"]]

        def read(self):
            return self.a

    class Writer(object):
        def write(self, a):
            self.a = a

        def write_geojson_features(self, fc):
            self.fc = fc

    filled, flowdir, depths, accum = Writer(), Writer(), Writer(), Writer()
    pipe = DemTool(Reader(fx["dtm"]), filled, flowdir, depths, accum).process(keep_pipeline=True)
    assert np.array_equal(filled.a, fx["filled"]) and np.array_equal(flowdir.a, fx["flowdir_noflats"])
    assert np.array_equal(depths.a, fx["depths"]) and accum.a.max() == 11158

    def run(filter_fn, pipeline):
        labeled, pour, wsheds = Writer(), Writer(), Writer()
        BluespotTool(input_depths=Reader(fx["depths"]), input_flowdir=Reader(fx["flowdir_noflats"]),
                     input_bluespot_filter_function=filter_fn, input_accum=Reader(accum.a), output_labeled_raster=labeled,
                     output_pourpoints=pour, output_watersheds_raster=wsheds, pipeline=pipeline).process()
        return labeled.a, pour.fc, wsheds.a

    lab, fc, ws = run(lambda s: True, pipe)
    pipe.close()
    assert lab.max() == 523 and len(fc["features"]) == 524          # tests/test_commandline.py:43 (no filter)
    # the CLI test's filter 'area > 20.5 and maxdepth > 0.5 or volume > 2.5' (tests/test_commandline.py:15) keeps 486
    lab2, fc2, ws2 = run(lambda s: s["area"] > 20.5 and s["max"] > 0.5 or s["volume"] > 2.5, None)
    assert lab2.max() == 486                                          # tests/test_commandline.py:24
    p = fc2["features"][5]["properties"]
    assert set(p) >= {"bspot_id", "cell_row", "cell_col", "bspot_dmax", "bspot_area", "bspot_vol", "wshed_area", "bspot_fumm"}
    m = lab2 > 0
    assert np.array_equal(ws2[m], lab2[m])


# ---- the whole chain in ONE request: label/watershed run on a second stream next to no-flats/D8/accumulation ----------

@pytest.mark.parametrize("shape", [(700, 450), (1024, 1024), (1500, 1300)])
def test_whole_chain_in_one_request_matches_oracle(alg, shape):
    """mhip_ctx_run with every stage bit set takes the overlapped path of the stage DAG (api.hip): same bits as the
    oracle run stage by stage."""
    from malstroem_amd.pipeline import HydroPipeline
    dem = fbm(shape[0], shape[1], seed=7)
    with HydroPipeline(dem.shape) as pipe:
        pipe.upload("dem", dem)
        for _ in range(2):   # the second pass reuses streams, events and pooled buffers
            pipe.run("fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints")
        pipe.sync()
        got = {k: pipe.download(k) for k in ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds")}
        stats, counts, pour = pipe.stats(), pipe.watershed_counts(), pipe.pourpoints()
        nlabels = pipe.get_int("nlabels")

    filled = oracle.fill_terrain(dem)
    short, diag = oracle.minimum_safe_short_and_diag(dem)
    noflat = oracle.fill_terrain_no_flats(dem, short, diag)
    flowdir = oracle.terrain_flowdirection(noflat)
    accum = oracle.accumulated_flow(flowdir)
    dep = oracle.depths(filled, dem)
    lab, n = oracle.connected_components(dep)
    ws = lab.copy()
    oracle.watersheds_from_labels(flowdir, ws)   # in place, like the reference
    assert np.array_equal(got["filled"], filled) and np.array_equal(got["depths"], dep)
    assert np.array_equal(got["noflat"], noflat) and np.array_equal(got["flowdir"], flowdir)
    assert np.array_equal(got["accum"], accum)
    assert nlabels == n and np.array_equal(got["labels"], lab) and np.array_equal(got["watersheds"], ws)
    ost = oracle.label_stats(dep, lab, n)
    for f in ("min", "max", "count"):
        assert np.array_equal(stats[f], ost[f]), f
    assert_label_sums(stats["sum"], ost["sum"], dep, lab)
    assert np.array_equal(counts, oracle.label_count(ws))   # every label owns at least its own cells: max(ws) == n
    opour = oracle.label_max_index(accum, lab, n)
    for f in opour.dtype.names:
        assert np.array_equal(pour[f], opour[f]), f


@pytest.mark.parametrize("beta,seed,shape", [(2.0, 101, (1536, 1280)), (3.0, 102, (1300, 1700)), (1.2, 103, (2048, 2048)), (2.5, 104, (997, 1501))])
def test_fills_match_oracle_on_more_terrain(alg, beta, seed, shape):
    """More shapes / roughness for the two fill kernels (macro-tile f32 path incl. ragged macro tiles, single-tile f64
    path): smooth terrain (large lakes, many rounds) and rough terrain (many small pits)."""
    dem = fbm(shape[0], shape[1], beta=beta, seed=seed)
    filled = alg.fill.fill_terrain(dem)
    assert np.array_equal(filled, oracle.fill_terrain(dem))
    short, diag = alg.fill.minimum_safe_short_and_diag(dem)
    assert (short, diag) == oracle.minimum_safe_short_and_diag(dem)
    assert np.array_equal(alg.fill.fill_terrain_no_flats(dem, short, diag), oracle.fill_terrain_no_flats(dem, short, diag))
