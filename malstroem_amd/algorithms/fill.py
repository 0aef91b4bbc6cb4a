"""Depression filling on the GPU -- mirror of ``malstroem.algorithms.fill`` (reference fill.py).

Whole-stage functions run as HIP kernels (csrc/fill.hip) through the C-ABI; there is no CPU
implementation in this package.  The reference's per-sweep helpers ``_fill_terrain`` /
``_fill_terrain_no_flats`` (fill.py:20-99) have no counterpart: a device pass is a tile round, not a
raster sweep (SURVEY.md 8b explains why the patch points move to whole stages).
"""
import ctypes

import numpy as np

from .. import _lib
from .dtypes import DTYPE_DTM, DTYPE_FILL, DTYPE_FILLNOFLAT


def _dtm(dtm):
    a = np.asarray(dtm)
    if a.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % a.ndim)
    if a.dtype != DTYPE_DTM:
        # the Cython sweep takes float32 buffers only (_fill.pyx:30)
        raise ValueError("Buffer dtype mismatch, expected 'float32' but got '%s'" % a.dtype)
    return np.ascontiguousarray(a)


def fill_terrain(dtm, return_rounds=False):
    """Depressionless DEM: every cell gets a non-uphill path to the raster edge (fill.py:112-171).

    float32 2-D in, float32 out.  Nodata is not supported (all values are elevations).
    """
    dtm = _dtm(dtm)
    out = np.empty(dtm.shape, dtype=DTYPE_FILL)
    rounds = ctypes.c_int32(0)
    _lib.call("mhip_fill_f32", _lib.ptr(dtm), _lib.ptr(out), _lib.i64(dtm.shape[0]), _lib.i64(dtm.shape[1]),
              ctypes.byref(rounds))
    return (out, rounds.value) if return_rounds else out


def fill_terrain_no_flats(dtm, short=0, diag=0, return_rounds=False):
    """Depressionless DEM with a strictly downslope path from every cell (fill.py:174-232).

    ``short`` / ``diag``: minimum elevation step between edge / corner neighbours.  float64 out.
    """
    dtm = _dtm(dtm)
    out = np.empty(dtm.shape, dtype=DTYPE_FILLNOFLAT)
    rounds = ctypes.c_int32(0)
    _lib.call("mhip_fill_noflat_f64", _lib.ptr(dtm), _lib.ptr(out), _lib.i64(dtm.shape[0]), _lib.i64(dtm.shape[1]),
              ctypes.c_double(short), ctypes.c_double(diag), ctypes.byref(rounds))
    return (out, rounds.value) if return_rounds else out


def minimum_safe_short_and_diag(dem):
    """Smallest safe (short, diag) for ``fill_terrain_no_flats`` (fill.py:235-250)."""
    dem = np.ascontiguousarray(dem, dtype=DTYPE_DTM)
    short, diag = ctypes.c_double(0), ctypes.c_double(0)
    _lib.call("mhip_short_diag", _lib.ptr(dem), _lib.i64(dem.size), ctypes.byref(short), ctypes.byref(diag))
    return np.float64(short.value), np.float64(diag.value)


def bluespot_depths(filled, dem):
    """``filled - dem`` as float32 (reference dem.py:71)."""
    f = np.ascontiguousarray(filled, dtype=DTYPE_FILL)
    d = np.ascontiguousarray(dem, dtype=DTYPE_DTM)
    if f.shape != d.shape:
        raise ValueError("shape mismatch")
    out = np.empty(f.shape, dtype=DTYPE_FILL)
    _lib.call("mhip_depths_f32", _lib.ptr(f), _lib.ptr(d), _lib.ptr(out), _lib.i64(f.size))
    return out
