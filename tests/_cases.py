"""Shared inputs for the parity tests: committed goldens + the synthetic fBm recipe."""
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"

PYREF_CASES = ["fbm64", "fbm97x61", "fbm160x192_b3", "noise33x29", "steps48x40",
               "tiny4x4", "tiny5x3", "tiny3x7", "negative40"]


def fixtures():
    """The reference's own golden rasters (decoded from its tests/data/*.tif)."""
    return np.load(GOLDEN / "reference_fixtures.npz")


def pyref(name):
    """Outputs of the reference's unmodified pure-Python path (make_pyref_goldens.py)."""
    return np.load(GOLDEN / ("pyref_%s.npz" % name))


def fbm(h, w=None, beta=2.0, seed=42):
    """Spectral-synthesis fBm, identical bits on every box from NumPy alone."""
    w = h if w is None else w
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


def reference_vectors():
    """The reference's own vector fixtures (pourpoints / nodes / streams .json), decoded by extract_reference_vectors.py."""
    import gzip
    import json
    with gzip.open(GOLDEN / "reference_vectors.json.gz", "rt") as fh:
        return json.load(fh)


def d8_mul_vs_div_case():
    """3x3 surface on which the two reference D8 variants disagree at the centre cell.

    The U neighbour drops by fl(dd * 1/2**0.5), the UR neighbour by dd = 484.544921875: the Cython path
    (_flow.pyx:93-94,140 multiplies the diagonal drop by 1/2**0.5) finds the two drops EQUAL, so U (code 0) keeps the
    strict `>` comparison; the Python twin (flow.py:79 divides by sqrt(2)) rounds the diagonal one ulp higher and takes
    UR (code 1).  The build follows the Cython variant.
    """
    dd = 484.544921875
    inv = 1.0 / 2 ** 0.5
    assert dd / (2 ** 0.5) > dd * inv
    z = np.zeros((3, 3), dtype=np.float64)
    z[0, 1] = -(dd * inv)
    z[0, 2] = -dd
    return z, 0, 1   # surface, code(cython), code(python)


def assemble_reference_pourpoints(cell_area, pp, stats, wcounts):
    """Field-by-field restatement of the pour point record of reference bluespots.py:75-82 (test-side copy for the oracle)."""
    out = []
    for ix in range(len(pp)):
        out.append(dict(bspot_id=ix, cell_row=int(pp["row"][ix]), cell_col=int(pp["col"][ix]),
                        bspot_dmax=float(stats["max"][ix]), bspot_area=float(stats["count"][ix] * cell_area),
                        bspot_vol=float(stats["sum"][ix] * cell_area), wshed_area=float(wcounts[ix] * cell_area)))
    return out


def assert_label_sums(got_sum, ref_sum, data, labels):
    """label_stats `sum` parity: the reference adds float64(values) sequentially in raster order.
    Wherever that sequential sum equals the exactly rounded sum (math.fsum) no partial sum ever rounded, every summation
    order gives the same bits, and the device must match EXACTLY; only for the remaining labels (none on the fixtures:
    depths are multiples of a float32 ulp) the comparison falls back to 1e-12 relative."""
    import math
    d = np.asarray(data, dtype=np.float64).ravel()
    lab = np.asarray(labels).ravel()
    order = np.argsort(lab, kind="stable")
    bounds = np.searchsorted(lab[order], np.arange(len(ref_sum) + 1))
    inexact = 0
    for l in range(len(ref_sum)):
        exact = math.fsum(d[order[bounds[l]:bounds[l + 1]]])
        if exact == ref_sum[l]:
            assert got_sum[l] == ref_sum[l], (l, got_sum[l], ref_sum[l])
        else:
            inexact += 1
            assert abs(got_sum[l] - ref_sum[l]) <= 1e-12 * abs(ref_sum[l]), (l, got_sum[l], ref_sum[l])
    return inexact


def serpentine_flowdir(h, w):
    """One river through EVERY cell: down column 0, right, up column 1, right, ... -- it crosses every row seam w times and
    the accumulated flow is the position along the river.  (AGNPS codes: 0 up, 2 right, 4 down, reference flow.py:30-38.)"""
    fd = np.empty((h, w), np.uint8)
    fd[:, 0::2] = 4
    fd[:, 1::2] = 0
    fd[h - 1, 0::2] = 2
    fd[0, 1::2] = 2
    return fd


def random_flowdir(h, w, seed, p_none=0.05):
    """Flow directions drawn at random: flow cycles, confluences, sinks and paths that wander across the seams."""
    rng = np.random.default_rng(seed)
    fd = rng.integers(0, 8, size=(h, w)).astype(np.uint8)
    fd[rng.random((h, w)) < p_none] = 8
    return fd


def meander_flowdir(h, w, seed):
    """Mostly downhill-to-the-right with vertical wiggles: long paths that collect tributaries while crossing seams."""
    rng = np.random.default_rng(seed)
    fd = rng.choice(np.array([1, 2, 3, 0, 4], np.uint8), size=(h, w), p=[0.3, 0.2, 0.3, 0.1, 0.1])
    return fd.astype(np.uint8)


def zigzag_flowdir(h, w, seed):
    """Every cell flows to the right, straight or diagonally (codes 1, 2, 3 at random): acyclic, and a path crosses a row seam
    every few cells, bouncing back and forth between neighbouring bands."""
    rng = np.random.default_rng(seed)
    return rng.choice(np.array([1, 2, 3], np.uint8), size=(h, w), p=[0.4, 0.2, 0.4]).astype(np.uint8)
