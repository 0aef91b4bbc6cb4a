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
