"""CPU-only: pins the oracle (oracle/malstroem_oracle.c) against
  (1) the reference's own golden rasters and known-answer scalars
      (reference tests/test_raster_{fill,flowdir,label}.py), and
  (2) outputs of the reference's unmodified pure-Python algorithms.
"""
import numpy as np
import pytest
import scipy.ndimage

import oracle
import math

from _cases import (PYREF_CASES, assemble_reference_pourpoints, d8_mul_vs_div_case, fixtures, pyref,
                    reference_vectors)


@pytest.fixture(scope="module")
def fx():
    return fixtures()


# ---- (1) reference fixtures -------------------------------------------------------------

def test_fill_matches_reference_golden(fx):
    # reference tests/test_raster_fill.py:7-13,26-33
    filled = oracle.fill_terrain(fx["dtm"])
    assert np.array_equal(filled, fx["filled"])
    assert filled.max() == fx["dtm"].max()
    # cython (exclusive) bounds give the same raster here: tests/test_raster_fill.py:46-55
    assert np.array_equal(oracle.fill_terrain(fx["dtm"], inclusive=False), fx["filled"])


def test_short_diag_known_answer(fx):
    short, diag = oracle.minimum_safe_short_and_diag(fx["dtm"])
    assert short == 7.275957614183426e-12 and diag == 1.0289757937229989e-11
    assert abs(diag / short - 2 ** 0.5) < 1e-12  # tests/test_raster_fill.py:70-72


def test_fill_no_flats_matches_reference_golden(fx):
    # reference tests/test_raster_fill.py:16-23,36-43
    short, diag = oracle.minimum_safe_short_and_diag(fx["dtm"])
    fnf = oracle.fill_terrain_no_flats(fx["dtm"], short, diag)
    assert fnf.dtype == np.float64
    assert np.array_equal(fnf, fx["filled_no_flats"])
    assert fnf.max() <= fx["dtm"].max() + sum(fx["dtm"].shape) * diag


def test_depths_matches_reference_golden(fx):
    assert np.array_equal(oracle.depths(fx["filled"], fx["dtm"]), fx["depths"])


def test_flowdir_matches_reference_golden(fx):
    # reference tests/test_raster_flowdir.py:12-26
    for variant in ("cython", "python"):
        fd = oracle.terrain_flowdirection(fx["filled_no_flats"], variant=variant)
        assert fd.dtype == np.uint8
        assert np.array_equal(fd, fx["flowdir_noflats"])
    assert np.bincount(fd.ravel()).tolist() == [6317, 4998, 9351, 6004, 7616, 3474, 5469, 3771]


def test_flowdir_rejects_float32(fx):
    with pytest.raises(ValueError):
        oracle.terrain_flowdirection(fx["filled"])  # _flow.pyx:99 buffer dtype mismatch


def test_accumulated_flow_known_answers(fx):
    # reference tests/test_raster_flowdir.py:49-70
    acc = oracle.accumulated_flow(fx["flowdir_noflats"])
    assert acc.dtype == np.float64
    assert acc.min() >= 1 and acc.max() == 11158 and acc.sum() == 3578615


def test_watersheds_match_reference_golden(fx):
    # reference tests/test_raster_flowdir.py:73-140
    ws = fx["labelled"].copy()
    oracle.watersheds_from_labels(fx["flowdir_noflats"], ws, 0)
    assert np.array_equal(ws, fx["wsheds"])
    assert ws.sum() == 2337891 and ws.max() == fx["labelled"].max()
    m = fx["labelled"] > 0
    assert np.array_equal(ws[m], fx["labelled"][m])


def test_connected_components_known_answers(fx):
    # reference tests/test_raster_label.py:8-16
    lab, n = oracle.connected_components(fx["filled_no_flats"] - fx["filled"])
    assert lab.dtype == np.int32 and n == 525
    assert (lab == 0).sum() == 40029 and lab.sum() == 1561377
    ref, nref = scipy.ndimage.label(fx["filled_no_flats"] - fx["filled"], structure=np.ones((3, 3)))
    assert nref == n and np.array_equal(ref, lab)


def test_label_stats_known_answers(fx):
    # reference tests/test_raster_label.py:19-35
    lab = fx["labelled"]
    st = oracle.label_stats(fx["depths"], lab)
    assert len(st) == lab.max() + 1
    assert st["count"].sum() == lab.size
    assert st["min"].min() == fx["depths"].min() and st["max"].max() == fx["depths"].max()
    big = int(np.argmax(st["count"][1:])) + 1
    sel = fx["depths"][lab == big]
    assert st[big]["min"] == sel.min() and st[big]["max"] == sel.max() and st[big]["count"] == sel.size
    np.testing.assert_almost_equal(st[big]["sum"], sel.astype(np.float64).sum())


def test_pourpoints_match_reference_vector_fixture(fx):
    """reference tests/data/pourpoints.json (tests/data/fixtures.py:13-15): all 105 pour point records are
    label_min_index(no-flats, labelled) + label_stats(depths, labelled) + label_count(wsheds) (bluespots.py:49-88,195-206)."""
    ref = reference_vectors()["pourpoints"]
    lab = fx["labelled"]
    gt = fx["geotransform"]
    cell_area = abs(gt[1]) * abs(gt[5])
    pp = oracle.label_min_index(fx["filled_no_flats"], lab)
    st = oracle.label_stats(fx["depths"], lab)
    wc = oracle.label_count(fx["wsheds"])
    mine = assemble_reference_pourpoints(cell_area, pp, st, wc)
    assert len(mine) == len(ref) == 105
    for m, r in zip(mine, ref):
        for k, v in m.items():
            assert r["properties"][k] == v, (m["bspot_id"], k)


def test_label_stats_sum_is_exact_on_the_fixtures(fx):
    """SURVEY 8a row S guard: the reference's sequential float64 sum equals the exactly rounded sum (math.fsum) for
    every label, so ANY summation order (the device's float64 atomics) reproduces the reference bits on these inputs."""
    st = oracle.label_stats(fx["depths"], fx["labelled"])
    d, lab = fx["depths"].astype(np.float64).ravel(), fx["labelled"].ravel()
    order = np.argsort(lab, kind="stable")
    bounds = np.searchsorted(lab[order], np.arange(len(st) + 1))
    for l in range(len(st)):
        assert st["sum"][l] == math.fsum(d[order[bounds[l]:bounds[l + 1]]]), l


def test_d8_follows_the_cython_variant_where_the_two_differ():
    z, code_cython, code_python = d8_mul_vs_div_case()
    assert oracle.terrain_flowdirection(z, variant="cython")[1, 1] == code_cython
    assert oracle.terrain_flowdirection(z, variant="python")[1, 1] == code_python
    assert code_cython != code_python


# ---- (2) pure-Python reference outputs ----------------------------------------------------

@pytest.mark.parametrize("name", PYREF_CASES)
def test_oracle_equals_python_reference(name):
    g = pyref(name)
    dem = g["dem"]
    filled = oracle.fill_terrain(dem)
    assert np.array_equal(filled, g["filled"])
    assert np.array_equal(oracle.depths(filled, dem), g["depths"])
    short, diag = oracle.minimum_safe_short_and_diag(dem)
    assert [short, diag] == g["short_diag"].tolist()
    fnf = oracle.fill_terrain_no_flats(dem, short, diag)
    assert np.array_equal(fnf, g["filled_no_flats"])
    # the python twin divides the diagonal drop by sqrt(2); the cython path multiplies
    assert np.array_equal(oracle.terrain_flowdirection(fnf, variant="python"), g["flowdir"])
    assert np.array_equal(oracle.terrain_flowdirection(fnf, False, variant="python"), g["flowdir_edges_nodir"])
    assert np.array_equal(oracle.terrain_flowdirection(filled.astype(np.float64), variant="python"),
                          g["flowdir_of_filled"])
    assert np.array_equal(oracle.accumulated_flow(g["flowdir"]), g["accum"])
    raw, n = oracle.connected_components(g["depths"])
    assert n == int(g["raw_nlabels"]) and np.array_equal(raw, g["raw_labels"])
    st = oracle.label_stats(g["depths"], raw)
    for f in ("min", "max", "sum", "count"):
        assert np.array_equal(st[f], g["raw_stats"][f]), f
    mask = oracle.keep_labels(raw, list(g["keepers"]))
    assert np.array_equal(mask, g["keep_mask"])
    lab, nl = oracle.connected_components(mask)
    assert nl == int(g["nlabels"]) and np.array_equal(lab, g["labeled"])
    ws = lab.copy()
    oracle.watersheds_from_labels(g["flowdir"], ws, 0)
    assert np.array_equal(ws, g["watersheds"])
    assert np.array_equal(oracle.label_count(ws), g["watershed_counts"])
    mi = oracle.label_min_index(fnf, lab, nl)
    ma = oracle.label_max_index(g["accum"], lab, nl)
    for f in ("value", "row", "col"):
        assert np.array_equal(mi[f], g["min_index"][f]), f
        assert np.array_equal(ma[f], g["max_index"][f]), f


@pytest.mark.parametrize("name", PYREF_CASES)
def test_d8_cython_variant_vs_python_variant(name):
    """Documents where multiply-by-1/sqrt2 (cython) and divide-by-sqrt2 (python) agree."""
    g = pyref(name)
    a = oracle.terrain_flowdirection(g["filled_no_flats"], variant="cython")
    b = oracle.terrain_flowdirection(g["filled_no_flats"], variant="python")
    assert np.array_equal(a, b)
