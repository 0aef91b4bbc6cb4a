"""Connected components and per-label reductions -- mirror of ``malstroem.algorithms.label`` (label.py).

All functions run as HIP kernels (csrc/ccl.hip, csrc/label_ops.hip); scipy is not used.
"""
import ctypes

import numpy as np

from .. import _lib
from .._lib import INDEX_DTYPE, STAT_DTYPE
from .dtypes import DTYPE_LABEL


def _labels(labelled):
    lab = np.asarray(labelled)
    if lab.dtype != DTYPE_LABEL:
        if lab.dtype.kind not in "iu":
            raise ValueError("integer label raster expected, got '%s'" % lab.dtype)
        if lab.size and (lab.max() > np.iinfo(np.int32).max or lab.min() < np.iinfo(np.int32).min):
            raise OverflowError("labels do not fit the int32 device representation")
    return np.ascontiguousarray(lab, dtype=DTYPE_LABEL)


def _nlabels(lab, nlabels):
    if not nlabels:   # same falsy test as the reference (_label.pyx:70-71)
        m = ctypes.c_int32(0)
        _lib.call("mhip_label_max", _lib.ptr(lab), _lib.i64(lab.size), ctypes.byref(m))
        nlabels = m.value
    return int(nlabels)


def connected_components(data):
    """8-connected components of ``data != 0`` numbered like ``scipy.ndimage.label`` (label.py:19-40).

    Returns ``(labels int32, nlabels)``; labels are ordered by each component's first raster pixel.
    """
    data = np.asarray(data)
    if data.ndim != 2:
        raise ValueError("2D array expected")
    lab = np.empty(data.shape, dtype=DTYPE_LABEL)
    n = ctypes.c_int64(0)
    H, W = _lib.i64(data.shape[0]), _lib.i64(data.shape[1])
    if data.dtype == np.float32:
        d = np.ascontiguousarray(data)
        _lib.call("mhip_ccl8_f32", _lib.ptr(d), _lib.ptr(lab), H, W, ctypes.byref(n))
    else:
        d = np.ascontiguousarray(data != 0).view(np.uint8)
        _lib.call("mhip_ccl8_u8", _lib.ptr(d), _lib.ptr(lab), H, W, ctypes.byref(n))
    return lab, int(n.value)


def label_stats(data, labelled, nlabels=None):
    """Per-label min, max, sum, count (label 0 included) as a record array (label.py:43-75)."""
    data = np.asarray(data)
    fn = "mhip_label_stats_f32"
    if data.dtype != np.float32:
        # the reference's generic path (label.py:43-75) accumulates any raster into float64 record fields
        if data.dtype.kind not in "fiub":
            raise ValueError("numeric raster expected, got '%s'" % data.dtype)
        data, fn = data.astype(np.float64), "mhip_label_stats_f64"
    data = np.ascontiguousarray(data)
    lab = _labels(labelled)
    if lab.shape != data.shape:
        raise ValueError("shape mismatch")
    nlabels = _nlabels(lab, nlabels)
    rec = np.zeros(nlabels + 1, dtype=STAT_DTYPE)
    _lib.call(fn, _lib.ptr(data), _lib.ptr(lab), _lib.i64(lab.size), _lib.i64(nlabels), _lib.ptr(rec))
    return rec


def keep_labels(labelled, keep_label, background=0):
    """Boolean raster that is True where the cell's label is kept (label.py:78-98).

    Like the reference this sets ``keep_label[background] = False`` on the caller's object.
    """
    keep_label[background] = False
    keep = np.ascontiguousarray(np.array(keep_label).astype(bool)).view(np.uint8)
    lab = _labels(labelled)
    mask = np.empty(lab.shape, dtype=np.uint8)
    _lib.call("mhip_keep_mask", _lib.ptr(lab), _lib.ptr(keep), _lib.i64(keep.size - 1), _lib.i64(lab.size), _lib.ptr(mask))
    return mask.view(bool)


def _index(fn, data, labelled, nlabels):
    data = np.ascontiguousarray(data, dtype=np.float64)   # the generic reference path compares as float64 too
    lab = _labels(labelled)
    if lab.shape != data.shape or lab.ndim != 2:
        raise ValueError("2D rasters of equal shape expected")
    nlabels = _nlabels(lab, nlabels)
    rec = np.zeros(nlabels + 1, dtype=INDEX_DTYPE)
    _lib.call(fn, _lib.ptr(data), _lib.ptr(lab), _lib.i64(lab.shape[0]), _lib.i64(lab.shape[1]), _lib.i64(nlabels),
              _lib.ptr(rec))
    return rec


def label_min_index(data, labelled, nlabels=None):
    """Per-label minimum and its first (row, col) in raster order (label.py:101-132)."""
    return _index("mhip_label_argmin_f64", data, labelled, nlabels)


def label_max_index(data, labelled, nlabels=None):
    """Per-label maximum and its first (row, col) in raster order (label.py:135-166)."""
    return _index("mhip_label_argmax_f64", data, labelled, nlabels)


def label_count(labelled):
    """``np.bincount(labelled.ravel())`` (label.py:169-180)."""
    lab = _labels(labelled)
    if lab.size == 0:
        return np.zeros(0, dtype=np.int64)
    n = _nlabels(lab, None)
    if n < 0:
        raise ValueError("'list' argument must have no negative elements")
    out = np.zeros(n + 1, dtype=np.int64)
    _lib.call("mhip_label_count", _lib.ptr(lab), _lib.i64(lab.size), _lib.i64(n), _lib.ptr(out))
    return out
