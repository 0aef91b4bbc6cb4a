"""CPU-only: the property the pour points of the device pipeline rest on (common.hpp: PourLink).

label_max_index(accumulated_flow, labels) -- bluespots.py:195-206 -- is a pass over two rasters.  Accumulated flow grows
strictly along a flow path, so the largest value of a label sits on a CANDIDATE: a cell whose downstream cell does not carry the same
label (or which has none).  The watersheds' tile pass lists the candidates, the accumulation's final pass turns them into keys.  Here:
the records over the candidates alone equal the reference's records over all cells, on the reference's own rasters and on random
labels; and a flow cycle (cells that stay 0) is what breaks the property -- the case the kernels flag and hand to the general pass."""
import numpy as np

import oracle
from _cases import fbm, fixtures

DR = np.array([-1, -1, 0, 1, 1, 1, 0, -1, 0])
DC = np.array([0, 1, 1, 1, 0, -1, -1, -1, 0])


def candidates(fd, labels):
    h, w = fd.shape
    code = np.minimum(fd, 8).astype(np.int64)
    rr, cc = np.mgrid[0:h, 0:w]
    nr, nc = rr + DR[code], cc + DC[code]
    ok = (code < 8) & (nr >= 0) & (nr < h) & (nc >= 0) & (nc < w)
    down = np.where(ok, labels[np.clip(nr, 0, h - 1), np.clip(nc, 0, w - 1)].astype(np.int64), np.int64(-2 ** 62))   # (no downstream cell)
    return down != labels


def records_over(mask, accum, labels, n):
    """label_max_index restricted to the cells of `mask`: the first cell in raster order among the largest of a label"""
    out = np.zeros(n + 1, dtype=[("value", float), ("row", int), ("col", int)])
    out["value"], out["row"], out["col"] = -np.inf, -1, -1
    w = accum.shape[1]
    idx = np.flatnonzero(mask.ravel())
    lab, val = labels.ravel()[idx], accum.ravel()[idx]
    order = np.lexsort((idx, -val, lab))          # label, then value descending, then position
    first = np.ones(len(order), bool)
    first[1:] = lab[order][1:] != lab[order][:-1]
    sel = order[first]
    out["value"][lab[sel]] = val[sel]
    out["row"][lab[sel]] = idx[sel] // w
    out["col"][lab[sel]] = idx[sel] % w
    return out


def check(fd, labels):
    n = int(labels.max())
    accum = oracle.accumulated_flow(fd)
    want = oracle.label_max_index(accum, labels, n)
    cand = candidates(fd, labels)
    got = records_over(cand, accum, labels, n)
    for f in want.dtype.names:
        assert np.array_equal(got[f], want[f]), f
    return cand.mean()


def test_the_reference_rasters():
    fx = fixtures()
    frac = check(fx["flowdir_noflats"], fx["labelled"])
    assert frac < 0.3          # a fifth of the cells (mostly unlabelled cells next to a bluespot), not all of them


def test_random_labels_on_a_d8_surface():
    rng = np.random.default_rng(3)
    dem = fbm(200, 260, seed=21)
    s, d = oracle.minimum_safe_short_and_diag(dem)
    fd = oracle.terrain_flowdirection(oracle.fill_terrain_no_flats(dem, s, d))
    labels = np.zeros(dem.shape, np.int32)
    for _ in range(300):       # rectangles: neighbours with different labels, one label in several places
        r, c = int(rng.integers(0, 192)), int(rng.integers(0, 252))
        labels[r:r + int(rng.integers(1, 9)), c:c + int(rng.integers(1, 9))] = int(rng.integers(1, 90))
    check(fd, labels)
    check(fd, np.arange(dem.size, dtype=np.int32).reshape(dem.shape) % 7)      # no background at all


def test_a_flow_cycle_is_what_breaks_it():
    fd = np.full((6, 8), 2, np.uint8)              # everything flows to the right and out ...
    fd[3, 3], fd[3, 4] = 2, 6                      # ... but two cells flow into each other: they stay 0
    labels = np.zeros(fd.shape, np.int32)
    labels[3, 3:5] = 1                             # a label that is the cycle alone has no candidate cell
    accum = oracle.accumulated_flow(fd)
    assert accum[3, 3] == 0 and accum[3, 4] == 0
    assert not candidates(fd, labels)[3, 3:5].any()
    assert oracle.label_max_index(accum, labels, 1)[1]["row"] == 3      # the reference's record: (0, first cell)
