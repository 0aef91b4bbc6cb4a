"""GPU: BASELINE configs[3] / [4] as far as one MI355X allows -- row bands of 65536-column rasters.

  * 16384 x 65536 (1.07 G cells, the band size of the 65536^2 run on 4 GPUs): the undivided chain on one context against
    the same raster cut into 4 bands (one band context each, threads over ThreadComm) -- bit for bit, raster by raster.
  * 65536 x 65536 (4.29 G cells > 2**31: no single context can hold it, int32 cell indices; BASELINE configs[3] at FULL size,
    the raster of `bench.py --gpus N`) as 4 bands of 16384 rows sharing the one GPU (91 % of its 288 GB): size-independent
    properties across the band seams -- the fill / no-flats fixed-point equations, D8, the accumulation balance, label
    numbering in scipy order across bands (new labels appear as running maximum + 1), watersheds following the flow, and
    pour-point records with GLOBAL rows that point at cells of their own bluespot holding the reported value.
The DEM is bench.py's two-octave 65536-wide surface (top rows of it).  These two tests move ~150 GB between host and device
and take a few minutes; everything else about bands is covered at small sizes in test_gpu_bands.py.
"""
import sys
import threading
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))

pytestmark = pytest.mark.gpu

W = 65536
RASTERS = ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds")
DR = (-1, -1, 0, 1, 1, 1, 0, -1)      # AGNPS codes 0..7: U UR R DR D DL L UL (reference flow.py:30-38)
DC = (0, 1, 1, 1, 0, -1, -1, -1)


@pytest.fixture(scope="module")
def source():
    from bench import DemSource
    return DemSource(W, 2.0)


def run_bands(source, H, nbands, work):
    """Runs the band chain on `nbands` threads; `work(p, comm, rec)` is called on every band's thread afterwards and its
    results are returned by rank.  The pipelines stay open (the caller closes them)."""
    from malstroem_amd.distributed import BandPipeline, ThreadComm
    out, pipes, err = [None] * nbands, [None] * nbands, []

    def body(comm):
        try:
            p = pipes[comm.rank] = BandPipeline(comm, (H, W), device=0)
            p.upload_dem(source.rows(p.row0, p.nrows))
            rec = p.run_chain(records=True, fetch_own=True)
            out[comm.rank] = work(p, comm, rec)
        except Exception as e:  # pragma: no cover
            import traceback
            traceback.print_exc()
            err.append(e)
            raise

    threads = [threading.Thread(target=body, args=(c,)) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(1500) for t in threads]
    assert not err, err
    return out, pipes


def test_16384x65536_undivided_equals_four_bands(source):
    from malstroem_amd.pipeline import HydroPipeline
    H = 16384
    bands, pipes = run_bands(source, H, 4, lambda p, comm, rec: dict(row0=p.row0, nrows=p.nrows, nlabels=p.nlabels))
    try:
        assert sum(b["nrows"] for b in bands) == H and all(b["nlabels"] == bands[0]["nlabels"] for b in bands)
        with HydroPipeline((H, W)) as pipe:       # the band contexts stay resident next to it: ~90 GB of the 288 GB
            for r0 in range(0, H, 2048):
                pipe.upload_rows("dem", r0, source.rows(r0, 2048))
            pipe.run("fill", "noflat", "flowdir", "accum", "label", "watershed", "pourpoints")
            pipe.sync()
            assert pipe.get_int("nlabels") == bands[0]["nlabels"] > 10 ** 6
            for k in RASTERS:
                for b, p in zip(bands, pipes):
                    got = p.download(k)
                    want = pipe.download_rows(k, b["row0"], b["nrows"])
                    assert got.shape == want.shape and np.array_equal(got, want), (k, b["row0"])
                    del got, want
    finally:
        for p in pipes:
            if p is not None:
                p.close()


def test_65536x65536_as_four_bands_on_one_gpu(source):
    H, nb = 65536, 4
    assert H * W > 2 ** 31

    def work(p, comm, rec):
        res = dict(row0=p.row0, nrows=p.nrows, nlabels=p.nlabels, label_range=p.label_range, short_diag=(p.short, p.diag))
        lab = p.download("labels")
        # scipy's numbering (order of first raster pixel): walking the band in raster order, a label above everything seen so
        # far -- starting from the last label of the bands above -- is always exactly the next integer
        rm = np.maximum.accumulate(np.concatenate([[p.label_range[0] - 1], lab.ravel()]))
        res["numbering_ok"] = bool(np.all(np.diff(rm) <= 1))
        res["max_label"] = int(rm[-1])
        del rm
        acc = p.download("accum")
        pour = rec["pour"]
        first = pour["first_label"]
        recs = pour["records"]
        rows, cols, vals = recs["row"], recs["col"], recs["value"]
        inside = (rows >= p.row0) & (rows < p.row0 + p.nrows)
        idx = np.flatnonzero(inside)[:200000]
        res["pour_checked"] = int(idx.size)
        res["pour_label_ok"] = bool(np.all(lab[rows[idx] - p.row0, cols[idx]] == first + idx))
        res["pour_value_ok"] = bool(np.all(acc[rows[idx] - p.row0, cols[idx]] == vals[idx]))
        res["pour_rows_global"] = bool(np.all((rows >= 0) & (rows < H)))
        res["stats_count_sum"] = int(rec["stats"]["records"]["count"].sum())
        res["ws_count_sum"] = int(rec["counts"]["records"].sum())
        res["background"] = (int(rec["stats"]["background"]["count"]), int(rec["counts"]["background"]))
        # seam rows for the cross-band equations: my first 2 and last 2 rows of the rasters the equations need
        for k in ("dem", "filled", "noflat", "flowdir", "accum", "labels", "watersheds"):
            a = acc if k == "accum" else lab if k == "labels" else p.download(k)
            res["head_" + k], res["tail_" + k] = a[:2].copy(), a[-2:].copy()
            del a
        return res

    bands, pipes = run_bands(source, H, nb, work)
    for p in pipes:
        p.close()
    n = bands[0]["nlabels"]
    assert all(b["nlabels"] == n for b in bands) and n > 10 ** 7
    # label numbering: bands number consecutive ranges in raster order; inside a band new labels appear as running max + 1
    assert bands[0]["label_range"][0] == 1 and bands[-1]["label_range"][1] == n
    for a, b in zip(bands[:-1], bands[1:]):
        assert b["label_range"][0] == a["label_range"][1] + 1
    assert all(b["numbering_ok"] for b in bands) and max(b["max_label"] for b in bands) == n
    # pour points: global rows, on a cell of their own bluespot, holding the reported accumulated flow
    assert all(b["pour_rows_global"] and b["pour_label_ok"] and b["pour_value_ok"] and b["pour_checked"] > 1000 for b in bands)
    # every cell is counted once: bluespot cells + background = all cells = watershed cells + unassigned
    assert sum(b["stats_count_sum"] for b in bands) + bands[0]["background"][0] == H * W
    assert sum(b["ws_count_sum"] for b in bands) + bands[0]["background"][1] == H * W
    short, diag = bands[0]["short_diag"]
    # the equations across every seam: rows (last 2 of the upper band, first 2 of the lower band) -> check the two middle rows
    for up, dn in zip(bands[:-1], bands[1:]):
        st = {k: np.concatenate([up["tail_" + k], dn["head_" + k]]) for k in ("dem", "filled", "noflat", "flowdir", "accum", "labels", "watersheds")}
        dem, F, G = st["dem"], st["filled"], st["noflat"]
        inner = (slice(1, 3), slice(1, W - 1))
        nb8 = [(dr, dc) for dr in (-1, 0, 1) for dc in (-1, 0, 1) if dr or dc]
        mn = np.minimum.reduce([F[1 + dr:3 + dr, 1 + dc:W - 1 + dc] for dr, dc in nb8])
        assert np.array_equal(F[inner], np.maximum(dem[inner], np.minimum(F[inner], mn)))                 # fill.py:20-62 fixed point
        assert np.array_equal(F[inner], np.maximum(dem[inner], mn))                                      # ... and the greatest one locally
        md = np.minimum.reduce([G[1 + dr:3 + dr, 1 + dc:W - 1 + dc] for dr, dc in nb8 if dr and dc]) + diag
        me = np.minimum.reduce([G[1 + dr:3 + dr, 1 + dc:W - 1 + dc] for dr, dc in nb8 if not (dr and dc)]) + short
        assert np.array_equal(G[inner], np.maximum(dem[inner].astype(np.float64), np.minimum(md, me)))   # _fill.pyx:107-117
        # D8 (_flow.pyx:98-176): first strict maximum of the drops, diagonals multiplied by 1/2**0.5
        z = G[inner]
        best, code = np.zeros_like(z), np.full(z.shape, 8, np.uint8)
        for k in range(8):
            dz = z - G[1 + DR[k]:3 + DR[k], 1 + DC[k]:W - 1 + DC[k]]
            if DR[k] and DC[k]:
                dz = dz * 0.7071067811865475
            take = dz > best
            best[take], code[take] = dz[take], k
        assert np.array_equal(st["flowdir"][inner], code)
        # accumulation balance (_flow.pyx:212-247): 1 + the accumulation of every neighbour that flows into the cell
        fd, acc = st["flowdir"], st["accum"]
        tot = np.ones((2, W - 2))
        for k in range(8):      # neighbour in direction k flows into me when its code is (k + 4) % 8
            nfd = fd[1 + DR[k]:3 + DR[k], 1 + DC[k]:W - 1 + DC[k]]
            tot += np.where(nfd == (k + 4) % 8, acc[1 + DR[k]:3 + DR[k], 1 + DC[k]:W - 1 + DC[k]], 0.0)
        assert np.array_equal(acc[inner], tot)
        # labels: 8-connected across the seam -- neighbouring bluespot cells carry one label
        lab = st["labels"]
        for dr, dc in nb8:
            a, c = lab[inner], lab[1 + dr:3 + dr, 1 + dc:W - 1 + dc]
            both = (a > 0) & (c > 0)
            assert np.array_equal(a[both], c[both])
        # watersheds follow the flow: an unlabelled cell carries the watershed of the cell it flows to (flow.py:367-412)
        ws = st["watersheds"]
        r_, c_ = np.nonzero((lab[inner] == 0) & (fd[inner] < 8))
        k_ = fd[inner][r_, c_].astype(int)
        down = ws[1 + r_ + np.take(DR, k_), 1 + c_ + np.take(DC, k_)]
        assert np.array_equal(ws[inner][r_, c_], down)
        assert np.array_equal(ws[inner][lab[inner] > 0], lab[inner][lab[inner] > 0])
