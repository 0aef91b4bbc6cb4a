"""Stream network between pour points -- mirror of ``malstroem.algorithms.net`` (reference net.py).

The raster part -- walking downstream from every pour point until another bluespot is met
(``next_downstream_label``, net.py:142-169, called once per bluespot by ``pourpoint_network`` /
``geometric_pourpoint_network``) -- runs as ONE batched HIP kernel over all pour points (csrc/trace.hip), on host
rasters or on the rasters a ``HydroPipeline`` already holds.  The graph surgery that inserts junction nodes where
streams merge (net.py:43-139) works on a few path lists and stays on the host, like in the reference.
"""
import ctypes
from collections import OrderedDict

import numpy as np

from .. import _lib
from .dtypes import DTYPE_FLOWDIR, DTYPE_LABEL

__all__ = ["next_downstream_label", "trace_downstream_labels", "pourpoint_network", "geometric_pourpoint_network", "network_from_walks"]


def _pourpoint_enumerator(pour_points):
    """(id, (row, col)) of every pour point: GeoJSON-like features (id = properties.bspot_id) or plain cells (id = position)."""
    for pid, pp in enumerate(pour_points):
        if isinstance(pp, dict) and 'properties' in pp:
            pid = pp['properties']['bspot_id']
            pp = (pp['properties']['cell_row'], pp['properties']['cell_col'])
        yield pid, (int(pp[0]), int(pp[1]))


def trace_downstream_labels(flowdir, labeled, cells, background_label=None, geometry=False, pipeline=None):
    """``next_downstream_label`` for many cells in one device pass.

    Returns ``(labels, geoms)``: ``labels[i]`` is an int or None, ``geoms[i]`` the list of (row, col) cells walked (empty
    lists unless ``geometry``).  ``pipeline``: a ``HydroPipeline`` whose resident flow directions and labels are used
    instead of the two host rasters (which may then be None)."""
    cells = np.ascontiguousarray(np.asarray(list(cells), dtype=np.int64).reshape(-1, 2))
    n = cells.shape[0]
    use_bg = background_label is not None
    bg = int(background_label) if use_bg else 0
    lab_out = np.zeros(n, dtype=np.int32)
    found = np.zeros(n, dtype=np.int32)
    lens = np.zeros(n, dtype=np.int64)
    if pipeline is None:
        fd = np.asarray(flowdir)
        lab = np.asarray(labeled)
        if fd.dtype != DTYPE_FLOWDIR:
            raise ValueError("Buffer dtype mismatch, expected 'uint8' but got '%s'" % fd.dtype)
        if lab.dtype != DTYPE_LABEL:
            if lab.dtype.kind not in "iu" or (lab.size and (lab.max() > np.iinfo(np.int32).max or lab.min() < np.iinfo(np.int32).min)):
                raise OverflowError("labels do not fit the int32 device representation")
        fd, lab = np.ascontiguousarray(fd), np.ascontiguousarray(lab, dtype=DTYPE_LABEL)
        if fd.shape != lab.shape or fd.ndim != 2:
            raise ValueError("2D rasters of equal shape expected")
        H, W = fd.shape

        def call(offsets, out_cells):
            _lib.call("mhip_trace_downstream_i32", _lib.ptr(fd), _lib.ptr(lab), _lib.i64(H), _lib.i64(W), _lib.ptr(cells), _lib.i64(n),
                      int(use_bg), ctypes.c_int32(bg), _lib.ptr(lab_out), _lib.ptr(found), _lib.ptr(lens),
                      None if offsets is None else _lib.ptr(offsets), None if out_cells is None else _lib.ptr(out_cells))
    else:
        H, W = pipeline.shape

        def call(offsets, out_cells):
            _lib.call("mhip_ctx_trace_downstream", pipeline._ctx, _lib.ptr(cells), _lib.i64(n), int(use_bg), ctypes.c_int32(bg),
                      _lib.ptr(lab_out), _lib.ptr(found), _lib.ptr(lens),
                      None if offsets is None else _lib.ptr(offsets), None if out_cells is None else _lib.ptr(out_cells))
    call(None, None)
    labels = [int(l) if f else None for l, f in zip(lab_out, found)]
    geoms = [[] for _ in range(n)]
    if geometry and n:
        offsets = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        flat = np.zeros(max(int(offsets[-1]), 1), dtype=np.int64)
        call(offsets, flat)
        rows, cols = np.divmod(flat, W)
        for i in range(n):
            a, b = int(offsets[i]), int(offsets[i + 1])
            geoms[i] = list(zip(rows[a:b].tolist(), cols[a:b].tolist()))
    return labels, geoms


def next_downstream_label(flowdir, labeled, cell, background_label=None, geometry=False):
    """First label downstream of ``cell`` that differs from the cell's own (and from ``background_label``), and the path
    walked (a list of cells, empty unless ``geometry``) -- reference net.py:142-169."""
    labels, geoms = trace_downstream_labels(flowdir, labeled, [cell], background_label, geometry)
    return labels[0], geoms[0]


def pourpoint_network(flowdir, labeled, pour_points, background_label=None, pipeline=None):
    """Node list ``{id, downstream_id, nodetype='pourpoint', pix}``, one per pour point (net.py:172-192)."""
    ids_cells = list(_pourpoint_enumerator(pour_points))
    labels, _ = trace_downstream_labels(flowdir, labeled, [c for _, c in ids_cells], background_label, False, pipeline)
    return [dict(id=pid, downstream_id=down, nodetype='pourpoint', pix=tuple(cell)) for (pid, cell), down in zip(ids_cells, labels)]


# ---- junction nodes where streams merge (host graph surgery, reference net.py:43-139) --------------------------------------

def _common_flow_groups(nodes, shared=2):
    """Partition ``nodes`` (all draining to the same node) into groups whose paths coincide on at least their last ``shared``
    cells; a node joins the group of the first earlier node it shares that many cells with (net.py:43-66)."""
    if len(nodes) <= 1:
        return [list(nodes)]
    groups, rest = [], list(nodes)
    while rest:
        head = rest.pop(0)
        group = [head]
        if len(head['geometry']) > shared:
            anchor = head['geometry'][-shared]
            keep = []
            for other in rest:
                if len(other['geometry']) > shared and other['geometry'][-shared] == anchor:
                    group.append(other)
                else:
                    keep.append(other)
            rest = keep
        groups.append(group)
    return groups


def _insert_junction(group, junction_id):
    """The streams of ``group`` share their tail: cut it off into a new junction node they all flow to (net.py:69-116)."""
    downstream_id = group[0]['downstream_id']
    assert all(n['downstream_id'] == downstream_id for n in group), "nodes of a group drain to one node"
    paths = [list(n['geometry']) for n in group]
    tail = []
    while all(p[-1] == paths[0][-1] for p in paths):
        tail.append(paths[0][-1])
        for p in paths:
            del p[-1]
    tail.reverse()
    junction = dict(id=junction_id, downstream_id=downstream_id, nodetype='junction', pix=tuple(tail[0]), geometry=tail)
    for n, p in zip(group, paths):
        n['downstream_id'] = junction_id
        n['geometry'] = p + [junction['pix']]
    return junction


def _untangle(nodes, next_label, out):
    """Depth-first: a junction per group of merging streams, then the same again among the streams above it (net.py:119-139).
    Appends the final nodes to ``out`` in the reference's order and returns the next free label."""
    for group in _common_flow_groups(nodes, 2):
        if len(group) > 1:
            out.append(_insert_junction(group, next_label))
            next_label = _untangle(group, next_label + 1, out)
        else:
            out.append(group[0])
    return next_label


def geometric_pourpoint_network(flowdir, labeled_bluespots, pour_points, background_label=None, pipeline=None, max_label=None):
    """Pour point network with junction nodes between the pour points and the path of every stream (net.py:195-224).

    ``max_label``: largest bluespot label if the caller knows it (a resident pipeline does); else taken from the raster."""
    ids_cells = list(_pourpoint_enumerator(pour_points))
    labels, geoms = trace_downstream_labels(flowdir, labeled_bluespots, [c for _, c in ids_cells], background_label, True, pipeline)
    if max_label is None:
        max_label = pipeline.get_int("nlabels") if pipeline is not None else int(np.max(labeled_bluespots))
    return network_from_walks([i for i, _ in ids_cells], [c for _, c in ids_cells], labels, geoms, int(max_label) + 1)


def network_from_walks(ids, cells, labels, geoms, next_label):
    """The node list of ``geometric_pourpoint_network`` from walks that have already been made (``labels[i]`` / ``geoms[i]``: what
    ``next_downstream_label`` returned for pour point ``ids[i]`` at ``cells[i]``) -- the row-band path walks across the bands
    (BandPipeline.trace_downstream) and hands the result to the same junction surgery (net.py:195-224).  ``next_label``: the
    first free node id (largest bluespot label + 1)."""
    upstream = OrderedDict()
    for pid, cell, down, geom in zip(ids, cells, labels, geoms):
        upstream.setdefault(down, []).append(dict(id=pid, downstream_id=down, nodetype='pourpoint', pix=(int(cell[0]), int(cell[1])),
                                                  geometry=[(int(r), int(c)) for r, c in geom]))
    final = []
    next_label = int(next_label)
    for nodes in upstream.values():
        next_label = _untangle(nodes, next_label, final)
    return final
