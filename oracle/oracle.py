"""ctypes front of ``liboracle.so`` (oracle/malstroem_oracle.c).  Test infrastructure only.

Function names and argument meaning mirror ``malstroem.algorithms.{fill,flow,label}`` of the
reference so that the parity tests read like the reference's own tests.
"""
import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

STAT_DTYPE = np.dtype([("min", "<f8"), ("max", "<f8"), ("sum", "<f8"), ("count", "<i8")])
INDEX_DTYPE = np.dtype([("value", "<f8"), ("row", "<i8"), ("col", "<i8")])

__all__ = [
    "build", "fill_terrain", "fill_terrain_no_flats", "minimum_safe_short_and_diag",
    "terrain_flowdirection", "accumulated_flow", "connected_components", "label_stats",
    "label_min_index", "label_max_index", "label_count", "keep_labels", "watersheds_from_labels",
    "depths", "STAT_DTYPE", "INDEX_DTYPE", "next_downstream_label", "rain_event",
]


def build(force=False):
    """Compile liboracle.so with gcc (seconds)."""
    so = _HERE / "liboracle.so"
    src = _HERE / "malstroem_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(str(build()))
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _check(rc, what):
    if rc != 0:
        raise ValueError("oracle %s failed with code %d" % (what, rc))


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    if a.ndim != 2:
        raise ValueError("2D array expected")
    return a


def fill_terrain(dtm, inclusive=True, return_sweeps=False):
    dtm = _c(dtm, np.float32)
    out = np.empty_like(dtm)
    n = ctypes.c_int32(0)
    _check(_lib().orc_fill_f32(_p(dtm), _p(out), ctypes.c_int64(dtm.shape[0]), ctypes.c_int64(dtm.shape[1]),
                               ctypes.c_int(int(inclusive)), ctypes.byref(n)), "fill_f32")
    return (out, n.value) if return_sweeps else out


def fill_terrain_no_flats(dtm, short=0.0, diag=0.0, inclusive=True, return_sweeps=False):
    dtm = _c(dtm, np.float32)
    out = np.empty(dtm.shape, dtype=np.float64)
    n = ctypes.c_int32(0)
    _check(_lib().orc_fill_noflat_f64(_p(dtm), _p(out), ctypes.c_int64(dtm.shape[0]), ctypes.c_int64(dtm.shape[1]),
                                      ctypes.c_double(short), ctypes.c_double(diag), ctypes.c_int(int(inclusive)),
                                      ctypes.byref(n)), "fill_noflat_f64")
    return (out, n.value) if return_sweeps else out


def minimum_safe_short_and_diag(dem):
    dem = np.ascontiguousarray(dem, dtype=np.float32)
    s, d = ctypes.c_double(0), ctypes.c_double(0)
    _check(_lib().orc_short_diag(_p(dem), ctypes.c_int64(dem.size), ctypes.byref(s), ctypes.byref(d)), "short_diag")
    return s.value, d.value


def terrain_flowdirection(terrain, edges_flow_outward=True, variant="cython"):
    if np.asarray(terrain).dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'float64'")  # _flow.pyx:99 accepts f64 only
    z = _c(terrain, np.float64)
    out = np.empty(z.shape, dtype=np.uint8)
    _check(_lib().orc_d8_f64(_p(z), _p(out), ctypes.c_int64(z.shape[0]), ctypes.c_int64(z.shape[1]),
                             ctypes.c_int(int(edges_flow_outward)), ctypes.c_int(0 if variant == "cython" else 1)), "d8")
    return out


def accumulated_flow(flowdir):
    fd = _c(flowdir, np.uint8)
    out = np.empty(fd.shape, dtype=np.float64)
    _check(_lib().orc_accum(_p(fd), _p(out), ctypes.c_int64(fd.shape[0]), ctypes.c_int64(fd.shape[1])), "accum")
    return out


def connected_components(data):
    data = np.asarray(data)
    if data.ndim != 2:
        raise ValueError("2D array expected")
    lab = np.empty(data.shape, dtype=np.int32)
    n = ctypes.c_int64(0)
    H, W = ctypes.c_int64(data.shape[0]), ctypes.c_int64(data.shape[1])
    if data.dtype == np.float32:
        d = np.ascontiguousarray(data)
        _check(_lib().orc_ccl8_f32(_p(d), _p(lab), H, W, ctypes.byref(n)), "ccl8_f32")
    else:
        d = np.ascontiguousarray(data != 0).view(np.uint8)
        _check(_lib().orc_ccl8_u8(_p(d), _p(lab), H, W, ctypes.byref(n)), "ccl8_u8")
    return lab, int(n.value)


def label_stats(data, labelled, nlabels=None):
    data = _c(data, np.float32)
    lab = _c(labelled, np.int32)
    if not nlabels:
        nlabels = int(lab.max())
    rec = np.zeros(nlabels + 1, dtype=STAT_DTYPE)
    _check(_lib().orc_label_stats_f32(_p(data), _p(lab), ctypes.c_int64(lab.size), ctypes.c_int64(nlabels), _p(rec)), "label_stats")
    return rec


def _index(fn, data, labelled, nlabels):
    data = _c(data, np.float64)
    lab = _c(labelled, np.int32)
    if not nlabels:
        nlabels = int(lab.max())
    rec = np.zeros(nlabels + 1, dtype=INDEX_DTYPE)
    _check(fn(_p(data), _p(lab), ctypes.c_int64(lab.shape[0]), ctypes.c_int64(lab.shape[1]), ctypes.c_int64(nlabels), _p(rec)), "label_index")
    return rec


def label_min_index(data, labelled, nlabels=None):
    return _index(_lib().orc_label_min_index_f64, data, labelled, nlabels)


def label_max_index(data, labelled, nlabels=None):
    return _index(_lib().orc_label_max_index_f64, data, labelled, nlabels)


def label_count(labelled):
    lab = np.ascontiguousarray(labelled, dtype=np.int32)
    n = int(lab.max())
    out = np.zeros(n + 1, dtype=np.int64)
    _check(_lib().orc_label_count(_p(lab), ctypes.c_int64(lab.size), ctypes.c_int64(n), _p(out)), "label_count")
    return out


def keep_labels(labelled, keep_label, background=0):
    lab = np.ascontiguousarray(labelled, dtype=np.int32)
    keep_label[background] = False
    keep = np.ascontiguousarray(np.array(keep_label).astype(bool)).view(np.uint8).copy()
    mask = np.empty(lab.shape, dtype=np.uint8)
    _check(_lib().orc_keep_labels(_p(lab), ctypes.c_int64(lab.size), _p(keep), ctypes.c_int64(keep.size), _p(mask)), "keep_labels")
    return mask.view(bool)


def watersheds_from_labels(flowdir, labelled, unassigned=0):
    """In place on ``labelled`` (int32, C-contiguous), like the reference."""
    fd = _c(flowdir, np.uint8)
    if labelled.dtype != np.int32 or not labelled.flags.c_contiguous:
        raise ValueError("int32 C-contiguous labelled expected")
    _check(_lib().orc_watersheds_i32(_p(fd), _p(labelled), ctypes.c_int64(fd.shape[0]), ctypes.c_int64(fd.shape[1]),
                                     ctypes.c_int32(unassigned)), "watersheds")


def depths(filled, dem):
    f = np.ascontiguousarray(filled, dtype=np.float32)
    d = np.ascontiguousarray(dem, dtype=np.float32)
    out = np.empty_like(f)
    _check(_lib().orc_depths_f32(_p(f), _p(d), _p(out), ctypes.c_int64(f.size)), "depths")
    return out


# ---- stream network (test infrastructure like everything in this package) -------------------------------------------------

def next_downstream_label(flowdir, labeled, cell, background_label=None, max_steps=None):
    """Plain-Python restatement of reference net.py:142-169 on top of flow.trace_downstream (flow.py:286-301): walk from
    ``cell`` along the flow directions; -> (first label that differs from the start cell's and from the background, or None;
    list of walked cells).  ``max_steps`` cuts flow cycles (the reference never returns from one)."""
    deltas = ((-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1))
    H, W = flowdir.shape
    r, c = int(cell[0]), int(cell[1])
    geom = []
    if not (0 <= r < H and 0 <= c < W):
        return None, geom
    src = labeled[r, c]
    cap = H * W if max_steps is None else max_steps
    while len(geom) < cap:
        geom.append((r, c))
        lbl = labeled[r, c]
        if lbl != src and (background_label is None or lbl != background_label):
            return int(lbl), geom
        k = int(flowdir[r, c])
        if k > 7:
            break
        r, c = r + deltas[k][0], c + deltas[k][1]
        if not (0 <= r < H and 0 <= c < W):
            break
    return None, geom


def rain_event(nodes, mmrain):
    """Plain-Python restatement of reference network.py:75-129: leaf-to-root fill / spill on the node forest.
    ``nodes``: dicts with nodeid, dstrnodeid, wshed_area, bspot_vol.  -> list of event dicts in the reference's order."""
    index, upstream, roots = {}, {}, []
    for n in nodes:
        index[n['nodeid']] = n
        upstream.setdefault(n['dstrnodeid'], []).append(n['nodeid'])
        if n['dstrnodeid'] is None:
            roots.append(n['nodeid'])
    values = {}
    for root in roots:
        order, stack = [], [root]
        while stack:
            x = stack.pop()
            order.append(x)
            stack.extend(upstream.get(x, []))
        for x in reversed(order):
            node = index[x]
            rainv = float(node['wshed_area']) * mmrain * 0.001
            capacity = float(node['bspot_vol'])
            inflow = sum([values[u]['spillv'] for u in upstream.get(x, [])]) if upstream.get(x) else 0.0
            total = rainv + inflow
            values[x] = dict(nodeid=x, rainv=rainv, spillv=max(0, total - capacity), v=min(total, capacity),
                             pctv=None if not capacity else 100.0 * min(total, capacity) / capacity)
    return list(values.values())
