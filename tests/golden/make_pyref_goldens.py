#!/usr/bin/env python3
"""Generate golden input/output vectors by running the reference's UNMODIFIED pure-Python
algorithms (``malstroem.algorithms.{fill,flow,label}`` imported from ``/root/reference``).

Only data (inputs + expected outputs) is written: ``tests/golden/pyref_<case>.npz``.
Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_pyref_goldens.py

The Cython ``speedups`` do not compile against this image's NumPy 2 / Cython 3 without
source edits, so ``speedups.available`` is False here and every function below is the
pure-Python twin, which the reference's own tests require to equal the Cython path
(``tests/test_raster_fill.py:46-67``, ``tests/test_raster_flowdir.py:12-26,49-70``,
``tests/test_raster_label.py:37-89``).  Two documented differences between the twins:
  * D8 diagonal drop: python divides by sqrt(2) (flow.py:79), cython multiplies by
    1/2**0.5 (_flow.pyx:93-94,140).  Both variants are stored when they differ.
  * fill sweeps: python bounds are inclusive (fill.py:27-28), i.e. the true fixed point.
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

REF = os.environ.get("MALSTROEM_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from malstroem.algorithms import fill, flow, label, speedups  # noqa: E402

assert not speedups.enabled, "goldens are defined on the pure-Python twin"

OUT = Path(__file__).resolve().parent


def fbm(h, w, beta=2.0, seed=42):
    """Spectral-synthesis fBm, SURVEY.md 8(d) recipe generalised to h x w."""
    rng = np.random.default_rng(seed)
    kx = np.fft.fftfreq(h)[:, None]
    ky = np.fft.rfftfreq(w)[None, :]
    k = np.hypot(kx, ky)
    k[0, 0] = 1
    amp = k ** (-(beta + 1) / 2)
    amp[0, 0] = 0
    phase = rng.normal(size=amp.shape) + 1j * rng.normal(size=amp.shape)
    z = np.fft.irfft2(amp * phase, s=(h, w))
    z = (z - z.min()) / (z.max() - z.min()) * 100
    return z.astype(np.float32)


def cases():
    rng = np.random.default_rng(7)
    yield "fbm64", fbm(64, 64)
    yield "fbm97x61", fbm(97, 61, seed=3)
    yield "fbm160x192_b3", fbm(160, 192, beta=3.0, seed=5)
    yield "noise33x29", rng.random((33, 29)).astype(np.float32) * 10
    # plateaus / ties: coarse quantisation gives large flats and many equal drops
    yield "steps48x40", np.round(fbm(48, 40, seed=11) / 12.5).astype(np.float32)
    yield "tiny4x4", (rng.random((4, 4)) * 5).astype(np.float32)
    yield "tiny5x3", (rng.random((5, 3)) * 5).astype(np.float32)
    yield "tiny3x7", (rng.random((3, 7)) * 5).astype(np.float32)
    neg = fbm(40, 40, seed=13) - 50.0
    neg[10:20, 10:20] = -9999.0  # nodata plateau, reference tests/test_raster_fill.py:75-82
    yield "negative40", neg.astype(np.float32)


def run_case(name, dem):
    t0 = time.time()
    g = {"dem": dem}
    filled = fill.fill_terrain(dem)
    g["filled"] = filled
    depths = filled - dem
    g["depths"] = depths
    short, diag = fill.minimum_safe_short_and_diag(dem)
    g["short_diag"] = np.array([short, diag], dtype=np.float64)
    fnf = fill.fill_terrain_no_flats(dem, short, diag)
    g["filled_no_flats"] = fnf
    fd = flow.terrain_flowdirection(fnf, edges_flow_outward=True)
    g["flowdir"] = fd
    g["flowdir_edges_nodir"] = flow.terrain_flowdirection(fnf, edges_flow_outward=False)
    # D8 on a surface WITH flats (interior NODIR), python twin
    g["flowdir_of_filled"] = flow.terrain_flowdirection(filled.astype(np.float64), edges_flow_outward=True)
    acc = flow.accumulated_flow(fd)
    g["accum"] = acc
    raw, nraw = label.connected_components(depths)
    assert raw.dtype == np.int32
    g["raw_labels"] = raw
    g["raw_nlabels"] = np.array(nraw)
    raw_stats = label.label_stats(depths, raw)
    g["raw_stats"] = raw_stats
    keepers = [bool(s["count"] >= 3 and s["max"] > 0.01) for s in raw_stats]
    g["keepers"] = np.array(keepers, dtype=bool)
    mask = label.keep_labels(raw, keepers)
    g["keep_mask"] = mask
    labeled, nlab = label.connected_components(mask)
    g["labeled"] = labeled
    g["nlabels"] = np.array(nlab)
    g["stats"] = label.label_stats(depths, labeled)
    ws = np.copy(labeled)
    flow.watersheds_from_labels(fd, ws, unassigned=0)
    g["watersheds"] = ws
    g["watershed_counts"] = label.label_count(ws)
    g["min_index"] = label.label_min_index(fnf, labeled, nlab)
    g["max_index"] = label.label_max_index(acc, labeled, nlab)
    np.savez_compressed(OUT / ("pyref_%s.npz" % name), **g)
    print("%-16s %-9s raw=%d kept=%d  %.1fs" % (name, dem.shape, nraw, nlab, time.time() - t0))


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for name, dem in cases():
        if only and name not in only:
            continue
        run_case(name, dem)
