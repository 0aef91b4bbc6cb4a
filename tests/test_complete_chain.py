"""The reference's end-to-end known answers (reference tests/test_commandline.py:10-47): ``malstroem complete -r 10 -r 100``
on tests/data/dtm.tif gives max bluespot label 486 and 544 event features with the filter
``area > 20.5 and maxdepth > 0.5 or volume > 2.5``; 523 and 587 without.

CPU: the oracle chain (C restatement of the rasters, Python restatement of the walk) + the product's host-side junction
surgery reproduces both pairs -- this pins the oracle's walk / the surgery against the reference's own end-to-end numbers.
GPU: ``malstroem_amd.complete.process_all`` from a GeoTIFF of the reference's DEM through io.RasterReader -> DemTool ->
BluespotTool -> StreamTool -> RainTool, every tool on the resident device pipeline, files read back and compared with the
oracle chain (rasters bit for bit, every event value).
"""
import json
import os
from collections import OrderedDict

import numpy as np
import pytest

import oracle
from oracle import oracle as O
from _cases import fixtures
from malstroem_amd.algorithms import net
from malstroem_amd.complete import parse_filter

CLI_FILTER = 'area > 20.5 and maxdepth > 0.5 or volume > 2.5'     # tests/test_commandline.py:15
KNOWN = {CLI_FILTER: (486, 544), None: (523, 587)}               # tests/test_commandline.py:24,28,43,47
RAIN = [10, 100]


@pytest.fixture(scope="module")
def fx():
    return fixtures()


@pytest.fixture(scope="module")
def oracle_chain(fx):
    """Everything `complete` computes, by the oracle, for both filters."""
    dtm = fx["dtm"]
    gt = [float(v) for v in fx["geotransform"]]
    area = abs(gt[1]) * abs(gt[5])
    filled = oracle.fill_terrain(dtm)
    dep = oracle.depths(filled, dtm)
    short, diag = oracle.minimum_safe_short_and_diag(dtm)
    noflat = oracle.fill_terrain_no_flats(dtm, short, diag)
    fd = oracle.terrain_flowdirection(noflat)
    raw, nraw = oracle.connected_components(dep)
    raw_stats = oracle.label_stats(dep, raw, nraw)
    out = dict(filled=filled, depths=dep, flowdir=fd, cases={})
    for flt in KNOWN:
        fn = parse_filter(flt)
        keep = np.array([bool(fn(dict(min=s["min"], max=s["max"], sum=s["sum"], count=s["count"], volume=s["sum"] * area,
                                      area=s["count"] * area))) for s in raw_stats])       # bluespots.py:23-46
        keep[0] = False
        lab, n = oracle.connected_components(oracle.keep_labels(raw, keep))              # bluespots.py:165-170
        stats = oracle.label_stats(dep, lab, n)
        ws = lab.copy()
        oracle.watersheds_from_labels(fd, ws, 0)
        wcount = oracle.label_count(ws)
        pp = oracle.label_min_index(noflat, lab, n)                                        # bluespots.py:203-205 (no -accum)
        upstream = OrderedDict()
        for pid in range(n + 1):
            cell = (int(pp["row"][pid]), int(pp["col"][pid]))
            down, geom = O.next_downstream_label(fd, lab, cell, 0)                          # net.py:142-169
            upstream.setdefault(down, []).append(dict(id=pid, downstream_id=down, nodetype='pourpoint', pix=cell, geometry=geom))
        nodes, nxt = [], n + 1
        for group in upstream.values():
            nxt = net._untangle(group, nxt, nodes)                                          # net.py:195-224
        props = []
        for nd in nodes:                                                                    # streams.py:77-100
            p = dict(nodeid=nd["id"], dstrnodeid=nd["downstream_id"], bspot_area=0.0, bspot_vol=0.0, wshed_area=0.0)
            if nd["nodetype"] == "pourpoint":
                i = nd["id"]
                p.update(bspot_area=stats["count"][i] * area, bspot_vol=stats["sum"][i] * area, wshed_area=wcount[i] * area)
            props.append(p)
        events = {mm: {e["nodeid"]: e for e in O.rain_event(props, mm)} for mm in RAIN}     # network.py:75-129
        out["cases"][flt] = dict(labels=lab, n=n, ws=ws, nodes=nodes, events=events, pp=pp)
    return out


@pytest.mark.parametrize("flt", list(KNOWN))
def test_oracle_chain_reproduces_the_reference_end_to_end_answers(oracle_chain, flt):
    case = oracle_chain["cases"][flt]
    assert (case["n"], len(case["nodes"])) == KNOWN[flt]
    assert int(case["labels"].max()) == KNOWN[flt][0]
    assert all(len(case["events"][mm]) == KNOWN[flt][1] for mm in RAIN)


def test_parse_filter_follows_the_reference_vocabulary():
    f = parse_filter("area > 20.5 and (maxdepth > 0.05 or volume > 2.5)")                  # scripts/complete.py:35
    assert f(dict(area=21, max=0.06, volume=0)) and not f(dict(area=20, max=1, volume=9))
    assert parse_filter(None)(dict()) and parse_filter("")(dict())
    g = parse_filter(CLI_FILTER)
    assert g(dict(area=0, max=0, volume=3)) and not g(dict(area=30, max=0.4, volume=2))
    for bad in ("__import__('os')", "area > 2; volume", "count > 3", "area.real > 1", "area > 'x'", "area >"):
        with pytest.raises(Exception):
            parse_filter(bad)


def test_vector_layers_round_trip(tmp_path):
    from malstroem_amd.io import VectorReader, VectorWriter
    feats = [dict(id=i, geometry=dict(type="Point", coordinates=[0.1 * i, 1 / 3.0]),
                  properties=dict(nodeid=i, v=np.float64(i) / 7, n=np.int64(i), none=None)) for i in range(5)]
    w = VectorWriter('GeoJSON', str(tmp_path / "vec"), 'nodes', None, None, None)
    w.write_geojson_features(feats)
    back = VectorReader(str(tmp_path / "vec"), 'nodes').read_geojson_features()
    assert [f["properties"] for f in back] == [dict(nodeid=i, v=i / 7, n=i, none=None) for i in range(5)]
    assert back[3]["geometry"]["coordinates"] == [0.1 * 3, 1 / 3.0] and back[3]["id"] == 3
    w.write_geojson_features(dict(type="FeatureCollection", features=feats[:2]))            # a collection, like BluespotTool's
    assert len(VectorReader(w.filepath).read_geojson_features()) == 2
    with pytest.raises(NotImplementedError):
        VectorWriter('ESRI shapefile', str(tmp_path), 'x', None, None, None)


def check_outputs(outdir, res, fx, want, flt):
    from malstroem_amd.io import RasterReader, VectorReader
    assert res["nlabels"] == KNOWN[flt][0]
    for name, ref in (("filled", fx["filled"]), ("bs_depths", fx["depths"]), ("flowdir", fx["flowdir_noflats"]),
                      ("bluespots", want["labels"]), ("watersheds", want["ws"])):
        with RasterReader(str(outdir / (name + ".tif"))) as r:
            a = r.read()
            assert a.dtype == ref.dtype and np.array_equal(a, ref), name
    events = VectorReader(res["vector"], "events").read_geojson_features()
    assert len(events) == KNOWN[flt][1]                                                    # tests/test_commandline.py:28,47
    pps = VectorReader(res["vector"], "pourpoints").read_geojson_features()
    assert [p["properties"]["bspot_id"] for p in pps] == list(range(KNOWN[flt][0] + 1))
    assert [(p["properties"]["cell_row"], p["properties"]["cell_col"]) for p in pps] == list(zip(want["pp"]["row"].tolist(), want["pp"]["col"].tolist()))
    nodes = VectorReader(res["vector"], "nodes").read_geojson_features()
    assert [(f["properties"]["nodeid"], f["properties"]["dstrnodeid"], f["properties"]["nodetype"]) for f in nodes] == \
        [(n["id"], n["downstream_id"], n["nodetype"]) for n in want["nodes"]]
    streams = VectorReader(res["vector"], "streams").read_geojson_features()
    assert len(streams) == sum(1 for n in want["nodes"] if n["geometry"])
    for f in events:
        p = f["properties"]
        for mm in RAIN:
            e = want["events"][mm][p["nodeid"]]
            for k in ("rainv", "spillv", "v", "pctv"):
                assert p["%s_%g" % (k, mm)] == e[k], (p["nodeid"], k, mm)                   # rain.py:67,86-87


def run_band_complete(tmp_path, fx, flt, nbands, backend_factory=None):
    """`complete` on `nbands` row bands (threads over ThreadComm): -> (outdir, result of rank 0)"""
    import threading
    from malstroem_amd.complete import process_all
    from malstroem_amd.distributed import ThreadComm
    from malstroem_amd.io import RasterWriter
    src = str(tmp_path / "dtm.tif")
    RasterWriter(src, tuple(float(v) for v in fx["geotransform"]), None, nodata=-9999.0).write(fx["dtm"])
    outdir = tmp_path / ("out%d" % nbands)
    outdir.mkdir()
    res, err = [None] * nbands, []

    def work(comm):
        try:
            res[comm.rank] = process_all(src, str(outdir), RAIN, filter=flt, comm=comm, backend_factory=backend_factory)
        except Exception as e:      # pragma: no cover
            import traceback
            traceback.print_exc()
            err.append(e)

    threads = [threading.Thread(target=work, args=(c,), daemon=True) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(900) for t in threads]
    assert not err and not any(t.is_alive() for t in threads), err
    assert all(r is None for r in res[1:])
    return outdir, res[0]


@pytest.mark.parametrize("flt,nbands", [(CLI_FILTER, 2), (None, 3)])
def test_complete_on_row_bands_cpu_stand_in(tmp_path, fx, oracle_chain, flt, nbands):
    """BASELINE configs[4] in miniature, protocol only (CpuBand computes with the oracle): the filtered chain on 2 and 3 bands --
    filter on the owner ranks, pour points from merged band records, walkers handed over at the seams, junction surgery and rain
    events on rank 0 -- reproduces the reference's end-to-end answers and every value of the undivided oracle chain."""
    from _cpu_band import CpuBand
    outdir, res = run_band_complete(tmp_path, fx, flt, nbands, backend_factory=CpuBand)
    check_outputs(outdir, res, fx, oracle_chain["cases"][flt], flt)


@pytest.mark.gpu
@pytest.mark.parametrize("flt,nbands", [(CLI_FILTER, 2), (CLI_FILTER, 3), (None, 4)])
def test_complete_on_row_bands(tmp_path, fx, oracle_chain, flt, nbands):
    """the same on the device: one band context per thread on one MI355X"""
    outdir, res = run_band_complete(tmp_path, fx, flt, nbands)
    check_outputs(outdir, res, fx, oracle_chain["cases"][flt], flt)


@pytest.mark.gpu
@pytest.mark.parametrize("flt", list(KNOWN))
def test_complete_from_the_dem_file_to_the_rain_events(tmp_path, fx, oracle_chain, flt):
    from malstroem_amd.complete import process_all
    from malstroem_amd.io import RasterReader, RasterWriter, VectorReader
    gt = tuple(float(v) for v in fx["geotransform"])
    src = str(tmp_path / "dtm.tif")
    RasterWriter(src, gt, None, nodata=-9999.0).write(fx["dtm"])
    outdir = tmp_path / "out"
    outdir.mkdir()
    res = process_all(src, str(outdir), RAIN, filter=flt)
    want = oracle_chain["cases"][flt]
    assert res["nlabels"] == KNOWN[flt][0]
    for name, ref in (("filled", fx["filled"]), ("bs_depths", fx["depths"]), ("flowdir", fx["flowdir_noflats"]),
                      ("bluespots", want["labels"]), ("watersheds", want["ws"])):
        with RasterReader(str(outdir / (name + ".tif"))) as r:
            a = r.read()
            assert a.dtype == ref.dtype and np.array_equal(a, ref), name
    assert not os.path.exists(str(outdir / "accum.tif"))
    with RasterReader(str(outdir / "bluespots.tif")) as r:
        assert int(r.read().max()) == KNOWN[flt][0]                                        # tests/test_commandline.py:24,43
    events = VectorReader(res["vector"], "events").read_geojson_features()
    assert len(events) == KNOWN[flt][1]                                                    # tests/test_commandline.py:28,47
    pps = VectorReader(res["vector"], "pourpoints").read_geojson_features()
    assert len(pps) == KNOWN[flt][0] + 1
    assert [(p["properties"]["cell_row"], p["properties"]["cell_col"]) for p in pps] == list(zip(want["pp"]["row"].tolist(), want["pp"]["col"].tolist()))
    nodes = VectorReader(res["vector"], "nodes").read_geojson_features()
    assert [(f["properties"]["nodeid"], f["properties"]["dstrnodeid"], f["properties"]["nodetype"]) for f in nodes] == \
        [(n["id"], n["downstream_id"], n["nodetype"]) for n in want["nodes"]]
    streams = VectorReader(res["vector"], "streams").read_geojson_features()
    assert len(streams) == sum(1 for n in want["nodes"] if n["geometry"])
    for f in events:
        p = f["properties"]
        for mm in RAIN:
            e = want["events"][mm][p["nodeid"]]
            for k in ("rainv", "spillv", "v", "pctv"):
                assert p["%s_%g" % (k, mm)] == e[k], (p["nodeid"], k, mm)                   # rain.py:67,86-87


@pytest.mark.gpu
def test_complete_with_accum_puts_pour_points_at_the_flow_maximum(tmp_path, fx):
    """`-accum` (scripts/complete.py:33,68): accum.tif is written and the pour points sit at the maximum accumulated flow of
    their bluespot (bluespots.py:195-200) instead of the minimum of the no-flats surface."""
    from malstroem_amd.complete import process_all
    from malstroem_amd.io import RasterReader, RasterWriter, VectorReader
    src = str(tmp_path / "dtm.tif")
    RasterWriter(src, tuple(float(v) for v in fx["geotransform"]), None).write(fx["dtm"])
    outdir = tmp_path / "out"
    outdir.mkdir()
    res = process_all(src, str(outdir), [20], accum=True, filter=CLI_FILTER)
    with RasterReader(str(outdir / "accum.tif")) as r:
        acc = r.read()
    assert acc.max() == 11158 and acc.sum() == 3578615                                      # tests/test_raster_flowdir.py known answers
    with RasterReader(str(outdir / "bluespots.tif")) as r:
        lab = r.read()
    opp = oracle.label_max_index(acc, lab, int(lab.max()))
    pps = VectorReader(res["vector"], "pourpoints").read_geojson_features()
    assert [(p["properties"]["cell_row"], p["properties"]["cell_col"]) for p in pps] == list(zip(opp["row"].tolist(), opp["col"].tolist()))
    with pytest.raises(ValueError):
        process_all(src, str(outdir), [20])                                                 # outdir not empty (scripts/complete.py:45-47)


@pytest.mark.gpu
def test_complete_on_row_bands_equals_one_context_on_a_larger_terrain(tmp_path):
    """2048 x 1536 fBm, ~40 000 bluespots, a filter that drops most of them, accumulated flow on: `complete` on 4 row bands against
    `complete` on one context -- every raster equal, the same pour points, nodes, streams and event values (walkers cross the seams
    thousands of times here, bluespots and watersheds straddle them)."""
    import threading
    from _cases import fbm
    from malstroem_amd.complete import process_all
    from malstroem_amd.distributed import ThreadComm
    from malstroem_amd.io import RasterReader, RasterWriter, VectorReader
    dem = fbm(2048, 1536, beta=2.0, seed=77) * np.float32(0.5)
    src = str(tmp_path / "dem.tif")
    RasterWriter(src, (500000.0, 1.6, 0.0, 6200000.0, 0.0, -1.6), None).write(dem)
    flt = "area > 30 and maxdepth > 0.02 or volume > 4"
    one = tmp_path / "one"
    one.mkdir()
    r1 = process_all(src, str(one), [20, 60], accum=True, filter=flt)
    bands = tmp_path / "bands"
    bands.mkdir()
    res, err = [None] * 4, []
    # (512 rows per band: every rank writes its own tile rows of every raster -- nothing may be gathered to rank 0)
    from malstroem_amd.distributed import BandPipeline
    monkey_gather = BandPipeline.gather_rows

    def no_gather(self, *a, **k):
        raise AssertionError("a raster went through the rank-0 funnel")

    def work(comm):
        try:
            BandPipeline.gather_rows = no_gather
            res[comm.rank] = process_all(src, str(bands), [20, 60], accum=True, filter=flt, comm=comm)
        except Exception as e:      # pragma: no cover
            import traceback
            traceback.print_exc()
            err.append(e)

    threads = [threading.Thread(target=work, args=(c,), daemon=True) for c in ThreadComm.world(4)]
    [t.start() for t in threads]
    [t.join(900) for t in threads]
    BandPipeline.gather_rows = monkey_gather
    assert not err and not any(t.is_alive() for t in threads), err
    r4 = res[0]
    assert r4["nlabels"] == r1["nlabels"] > 100
    for name in ("filled", "bs_depths", "flowdir", "accum", "bluespots", "watersheds"):
        with RasterReader(str(one / (name + ".tif"))) as a, RasterReader(str(bands / (name + ".tif"))) as b:
            assert np.array_equal(a.read(), b.read()), name
    for layer in ("pourpoints", "nodes", "streams", "events"):
        fa = VectorReader(r1["vector"], layer).read_geojson_features()
        fb = VectorReader(r4["vector"], layer).read_geojson_features()
        assert len(fa) == len(fb) > 0, layer
        assert fa == fb, layer
