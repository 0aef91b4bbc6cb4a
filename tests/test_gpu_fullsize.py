"""GPU, BASELINE.json's full size (16384 x 16384 fBm, the bench workload): the oracle would need minutes per stage there,
so the chain is checked through size-independent properties that pin each output completely or almost completely:

  fill        fixed point of W = max(dem, min(W, 8 nbrs)) with border == dem, W >= dem, and idempotence fill(W) == W
              (a fixed point that is not the GREATEST one would have to differ from the oracle on the smaller parity cases
              that run the same kernels)
  no-flats    exact f64 fixed point of W = max(dem, min(W, min4(diag)+diag, min4(edge)+short)), W >= filled
  D8          recomputed with NumPy on a band of rows (same arithmetic: multiply by 0.7071067811865475, strict >)
  accum       accum == 1 + sum of accum over the upstream neighbours, for every cell (bincount over downstream indices)
  labels      foreground == (depths != 0); 8-neighbours that are both foreground carry the same label; labels 1..n each
              used; first raster occurrence increases with the label (scipy's numbering)
  watersheds  == labels on labelled cells; an unlabelled cell carries the value of its downstream cell (0 if it leaves)
  records     label_stats counts / min / max / sum, watershed counts, pour points = first arg-max of accum per label
"""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(1500)]

N = 16384
DR = (-1, -1, 0, 1, 1, 1, 0, -1)
DC = (0, 1, 1, 1, 0, -1, -1, -1)


@pytest.fixture(scope="module")
def run():
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    from bench import fbm
    from malstroem_amd.pipeline import HydroPipeline
    dem = fbm(N, beta=2.0, seed=42)
    out = {"dem": dem}
    with HydroPipeline(dem.shape) as pipe:
        pipe.upload("dem", dem)
        pipe.run("fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints")
        pipe.sync()
        for k in ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds"):
            out[k] = pipe.download(k)
        out["stats"], out["counts"], out["pour"] = pipe.stats(), pipe.watershed_counts(), pipe.pourpoints()
        out["nlabels"] = pipe.get_int("nlabels")
        out["short"], out["diag"] = pipe.get_float("short"), pipe.get_float("diag")
    with HydroPipeline(dem.shape) as pipe:   # idempotence: the filled surface is its own fill
        pipe.upload("dem", out["filled"])
        pipe.run("fill")
        out["refilled"] = pipe.download("filled")
    return out


def _nbr_min(w, which):
    """min over the neighbours `which` (direction indices) of the interior cells, as an (H-2, W-2) array"""
    H, W = w.shape
    m = None
    for k in which:
        v = w[1 + DR[k]:H - 1 + DR[k], 1 + DC[k]:W - 1 + DC[k]]
        m = v.copy() if m is None else np.minimum(m, v, out=m)
    return m


def test_fill_is_an_idempotent_fixed_point(run):
    dem, w = run["dem"], run["filled"]
    assert w.dtype == np.float32 and (w >= dem).all()
    for sl in (np.s_[0, :], np.s_[-1, :], np.s_[:, 0], np.s_[:, -1]):
        assert np.array_equal(w[sl], dem[sl])
    m = np.minimum(_nbr_min(w, range(8)), w[1:-1, 1:-1])
    assert np.array_equal(np.maximum(dem[1:-1, 1:-1], m), w[1:-1, 1:-1])
    assert np.array_equal(run["refilled"], w)
    assert np.array_equal(run["depths"], w - dem)


def test_noflats_is_an_exact_f64_fixed_point(run):
    dem, w, f = run["dem"], run["noflat"], run["filled"]
    short, diag = run["short"], run["diag"]
    maxval = np.float64(max(abs(np.float32(dem.max())), abs(np.float32(dem.min()))))
    assert short == (np.nextafter(maxval, np.inf) - maxval) * 1024 and diag == short * 2 ** 0.5
    assert w.dtype == np.float64 and (w >= f.astype(np.float64)).all()
    for sl in (np.s_[0, :], np.s_[-1, :], np.s_[:, 0], np.s_[:, -1]):
        assert np.array_equal(w[sl], dem[sl].astype(np.float64))
    cand = np.minimum(_nbr_min(w, (1, 3, 5, 7)) + diag, _nbr_min(w, (0, 2, 4, 6)) + short)
    np.minimum(cand, w[1:-1, 1:-1], out=cand)
    assert np.array_equal(np.maximum(dem[1:-1, 1:-1].astype(np.float64), cand), w[1:-1, 1:-1])
    assert float((w - f).max()) <= 1.01 * w.size * diag      # the seed bound of fill_noflat_dev


def test_d8_matches_numpy_on_a_band(run):
    z, fd = run["noflat"], run["flowdir"]
    assert fd.dtype == np.uint8 and fd.max() <= 7                       # no interior NODIR on a no-flats surface
    r0, r1 = 6000, 6512
    zz = z[r0 - 1:r1 + 1]
    best = np.zeros((r1 - r0, N - 2))
    code = np.full((r1 - r0, N - 2), 8, np.uint8)
    c = zz[1:-1, 1:-1]
    for k in range(8):
        dz = c - zz[1 + DR[k]:zz.shape[0] - 1 + DR[k], 1 + DC[k]:N - 1 + DC[k]]
        if k % 2:
            dz = dz * 0.7071067811865475
        take = dz > best
        best[take] = dz[take]
        code[take] = k
    assert np.array_equal(code, fd[r0:r1, 1:-1])
    assert (fd[0, 1:-1] == 0).all() and (fd[-1, 1:-1] == 4).all() and (fd[1:-1, 0] == 6).all() and (fd[1:-1, -1] == 2).all()
    assert (fd[0, 0], fd[0, -1], fd[-1, -1], fd[-1, 0]) == (7, 1, 3, 5)


def _downstream(fd):
    H, W = fd.shape
    rr, cc = np.divmod(np.arange(H * W, dtype=np.int64), W)
    dr = np.array(DR + (0,), np.int64)[fd.ravel()]
    dc = np.array(DC + (0,), np.int64)[fd.ravel()]
    nr, nc = rr + dr, cc + dc
    inside = (nr >= 0) & (nr < H) & (nc >= 0) & (nc < W) & (fd.ravel() <= 7)
    return np.where(inside, nr * W + nc, -1)


def test_accumulation_balances_at_every_cell(run):
    fd, acc = run["flowdir"], run["accum"]
    assert acc.dtype == np.float64 and acc.min() >= 1.0                 # acyclic: every cell resolves
    assert np.array_equal(acc, np.rint(acc))
    d = _downstream(fd)
    ok = d >= 0
    inflow = np.bincount(d[ok], weights=acc.ravel()[ok], minlength=acc.size)
    assert np.array_equal(acc.ravel(), inflow + 1.0)
    assert acc.ravel()[~ok].sum() == acc.size                           # everything leaves through the border cells
    run["down"] = d


def test_labels_are_the_8_connected_components_in_scipy_order(run):
    dep, lab, n = run["depths"], run["labels"], run["nlabels"]
    assert lab.dtype == np.int32 and np.array_equal(lab > 0, dep != 0) and lab.min() == 0 and lab.max() == n
    H, W = lab.shape
    for k in (2, 3, 4, 5):      # E, SE, S, SW cover every unordered neighbour pair once
        a = lab[max(0, -DR[k]):H - max(0, DR[k]), max(0, -DC[k]):W - max(0, DC[k])]
        b = lab[max(0, DR[k]):H + min(0, DR[k]) or None, max(0, DC[k]):W + min(0, DC[k]) or None]
        both = (a > 0) & (b > 0)
        assert np.array_equal(a[both], b[both])
    flat = lab.ravel()
    fg = np.flatnonzero(flat)
    vals, first = np.unique(flat[fg], return_index=True)
    assert np.array_equal(vals, np.arange(1, n + 1)) and (np.diff(fg[first]) > 0).all()
    # components that touch are one component: a label never has two disjoint... covered by the oracle parity cases;
    # here: the number of labels equals the number of cells whose W/NW/N/NE neighbours are all background or ... (roots)
    st = run["stats"]
    assert np.array_equal(st["count"], np.bincount(flat, minlength=n + 1))
    assert int(st["count"].sum()) == flat.size


def test_label_stats_records(run):
    dep, lab, n, st = run["depths"], run["labels"], run["nlabels"], run["stats"]
    import scipy.ndimage as ndi
    idx = np.arange(0, n + 1)
    assert np.array_equal(st["max"], ndi.maximum(dep, lab, idx).astype(np.float64))
    assert np.array_equal(st["min"], ndi.minimum(dep, lab, idx).astype(np.float64))
    sums = np.bincount(lab.ravel(), weights=dep.ravel().astype(np.float64), minlength=n + 1)
    assert np.allclose(st["sum"], sums, rtol=1e-12, atol=0)


def test_watersheds_follow_the_flow(run):
    lab, ws, fd = run["labels"], run["watersheds"], run["flowdir"]
    d = run.get("down")
    if d is None:
        d = _downstream(fd)
    w = ws.ravel()
    labelled = lab.ravel() > 0
    assert np.array_equal(w[labelled], lab.ravel()[labelled])
    free = ~labelled
    leaves = free & (d < 0)
    assert (w[leaves] == 0).all()
    inner = free & (d >= 0)
    assert np.array_equal(w[inner], w[d[inner]])
    assert np.array_equal(run["counts"], np.bincount(w, minlength=run["nlabels"] + 1))


def test_pour_points_are_first_argmax_of_accum(run):
    acc, lab, n, pp = run["accum"], run["labels"], run["nlabels"], run["pour"]
    import scipy.ndimage as ndi
    idx = np.arange(0, n + 1)
    assert np.array_equal(pp["value"], ndi.maximum(acc, lab, idx))
    assert np.array_equal(acc[pp["row"], pp["col"]], pp["value"]) and np.array_equal(lab[pp["row"], pp["col"]], idx)
    # FIRST raster position of the maximum: no earlier cell of the label carries the same value
    lin = pp["row"] * N + pp["col"]
    at_max = acc.ravel() == pp["value"][lab.ravel()]
    first = np.full(n + 1, acc.size, np.int64)
    pos = np.flatnonzero(at_max)
    np.minimum.at(first, lab.ravel()[pos], pos)
    assert np.array_equal(first, lin)


def test_headline_config_equals_the_oracle_cell_by_cell(run):
    """BASELINE configs[2] against the ORACLE itself, every cell of every raster and every record (the reference's own tests
    compare whole rasters: tests/test_raster_fill.py:46-67, test_raster_flowdir.py:49-70, test_raster_label.py:8-16).  The
    single-thread C restatement needs ~90 s for the 268 M cells; the two branches behind the plain fill run on two threads."""
    import threading
    import oracle
    dem = run["dem"]
    filled = oracle.fill_terrain(dem)
    assert np.array_equal(run["filled"], filled), "filled"
    short, diag = oracle.minimum_safe_short_and_diag(dem)
    assert (run["short"], run["diag"]) == (short, diag)
    side, errs = {}, []

    def label_branch():
        try:
            dep = oracle.depths(filled, dem)
            side["depths_ok"] = np.array_equal(run["depths"], dep)
            lab, n = oracle.connected_components(dep)
            side["lab"], side["n"] = lab, n
            side["stats"] = oracle.label_stats(dep, lab, n)
        except Exception as e:      # surfaced on the main thread
            errs.append(e)

    t = threading.Thread(target=label_branch)
    t.start()
    noflat = oracle.fill_terrain_no_flats(dem, short, diag)
    assert np.array_equal(run["noflat"], noflat), "noflat"
    fd = oracle.terrain_flowdirection(noflat)
    del noflat
    assert np.array_equal(run["flowdir"], fd), "flowdir"
    acc = oracle.accumulated_flow(fd)
    assert np.array_equal(run["accum"], acc), "accum"
    t.join()
    assert not errs, errs
    assert side["depths_ok"], "depths"
    lab, n = side["lab"], side["n"]
    assert n == run["nlabels"] and np.array_equal(run["labels"], lab), "labels"
    st = side["stats"]
    for f in ("min", "max", "count"):
        assert np.array_equal(run["stats"][f], st[f]), f
    # label_stats.sum is an order-dependent f64 sum in the reference (_label.pyx:91): exact wherever it equals the exactly rounded
    # sum (depths are multiples of a float32 ulp: everywhere on the fixtures), 1e-12 relative as the stated bar
    assert np.allclose(run["stats"]["sum"], st["sum"], rtol=1e-12, atol=0)
    assert (run["stats"]["sum"] == st["sum"]).mean() > 0.999
    pour = oracle.label_max_index(acc, lab, n)
    for f in ("value", "row", "col"):
        assert np.array_equal(run["pour"][f], pour[f]), f
    del acc
    ws = lab.copy()
    oracle.watersheds_from_labels(fd, ws, 0)
    assert np.array_equal(run["watersheds"], ws), "watersheds"
    assert np.array_equal(run["counts"], oracle.label_count(ws))
