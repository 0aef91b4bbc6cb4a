"""Flow direction, accumulation and watersheds -- mirror of ``malstroem.algorithms.flow`` (flow.py).

Raster-wide stages (``terrain_flowdirection``, ``accumulated_flow``, ``watersheds_from_labels``) run
as HIP kernels; the per-cell helpers used by the stream-network code (``upstream_cells``,
``trace_downstream`` ...) are tiny host utilities on NumPy arrays.
"""
import numpy as np

from .. import _lib
from ._raster_utils import cell_in_raster
from .dtypes import DTYPE_ACCUM, DTYPE_FILLNOFLAT, DTYPE_FLOWDIR

# AGNPS flow direction codes (reference flow.py:30-38).  Kernels depend on these exact values.
FLOWDIR_UP = 0
FLOWDIR_UP_RIGHT = 1
FLOWDIR_RIGHT = 2
FLOWDIR_DOWN_RIGHT = 3
FLOWDIR_DOWN = 4
FLOWDIR_DOWN_LEFT = 5
FLOWDIR_LEFT = 6
FLOWDIR_UP_LEFT = 7
FLOWDIR_NODIR = 8

_DELTAS = ((-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1))  # flow.py:186-209


def _terrain_flow(terrain):
    """D8 codes of the interior, NODIR on the border (reference _flow.pyx:98-176)."""
    return terrain_flowdirection(terrain, edges_flow_outward=False)


def terrain_flowdirection(terrain, edges_flow_outward=True):
    """Steepest-descent (D8) flow direction of every cell (flow.py:142-167).

    float64 surface only, like the Cython path (_flow.pyx:99).  Water never flows uphill nor between
    equal cells.  With ``edges_flow_outward`` border cells point off the raster, else they are NODIR.
    """
    z = np.asarray(terrain)
    if z.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % z.ndim)
    if z.dtype != DTYPE_FILLNOFLAT:
        raise ValueError("Buffer dtype mismatch, expected 'float64' but got '%s'" % z.dtype)
    z = np.ascontiguousarray(z)
    out = np.empty(z.shape, dtype=DTYPE_FLOWDIR)
    _lib.call("mhip_d8_f64", _lib.ptr(z), _lib.ptr(out), _lib.i64(z.shape[0]), _lib.i64(z.shape[1]),
              int(bool(edges_flow_outward)))
    return out


def set_edges_flow_outward(flowdir):
    """Force border cells to flow off the raster, in place (flow.py:118-139)."""
    maxr, maxc = flowdir.shape[0] - 1, flowdir.shape[1] - 1
    flowdir[0, :] = FLOWDIR_UP
    flowdir[maxr, :] = FLOWDIR_DOWN
    flowdir[:, 0] = FLOWDIR_LEFT
    flowdir[:, maxc] = FLOWDIR_RIGHT
    flowdir[0, 0] = FLOWDIR_UP_LEFT
    flowdir[0, maxc] = FLOWDIR_UP_RIGHT
    flowdir[maxr, 0] = FLOWDIR_DOWN_LEFT
    flowdir[maxr, maxc] = FLOWDIR_DOWN_RIGHT


def _flowdir(flowdir):
    fd = np.asarray(flowdir)
    if fd.ndim != 2:
        raise ValueError("Buffer has wrong number of dimensions (expected 2, got %d)" % fd.ndim)
    if fd.dtype != DTYPE_FLOWDIR:
        raise ValueError("Buffer dtype mismatch, expected 'uint8' but got '%s'" % fd.dtype)
    return np.ascontiguousarray(fd)


def accumulated_flow(flowdir):
    """Number of cells draining through each cell, itself included (flow.py:344-364); float64."""
    fd = _flowdir(flowdir)
    out = np.empty(fd.shape, dtype=DTYPE_ACCUM)
    _lib.call("mhip_accum", _lib.ptr(fd), _lib.ptr(out), _lib.i64(fd.shape[0]), _lib.i64(fd.shape[1]))
    return out


def watersheds_from_labels(flowdir, labelled, unassigned):
    """Grow every label upstream over the unassigned cells draining into it, IN PLACE (flow.py:398-412)."""
    fd = _flowdir(flowdir)
    if labelled.shape != fd.shape:
        raise ValueError("shape mismatch")
    work = labelled
    if labelled.dtype != np.int32 or not labelled.flags.c_contiguous:
        # int64 / other integer label rasters (the reference has i64 and generic variants, _flow.pyx:397-403)
        if labelled.size and (labelled.max() > np.iinfo(np.int32).max or labelled.min() < np.iinfo(np.int32).min):
            raise OverflowError("labels do not fit the int32 device representation")
        work = np.ascontiguousarray(labelled, dtype=np.int32)
    _lib.call("mhip_watersheds_i32", _lib.ptr(fd), _lib.ptr(work), _lib.i64(fd.shape[0]), _lib.i64(fd.shape[1]),
              int(unassigned))
    if work is not labelled:
        labelled[...] = work


# ---- per-cell host helpers (reference flow.py:170-301), used by stream tracing and tests --------------

def direction_to_delta(direction):
    """(row_delta, col_delta) of an AGNPS code; None for None / NODIR."""
    if direction is None or direction == FLOWDIR_NODIR:
        return None
    if 0 <= direction <= 7:
        return _DELTAS[int(direction)]
    raise Exception("Unknown flow direction code: {}".format(direction))


def cell_in_direction(cell, direction):
    delta = direction_to_delta(direction)
    return (cell[0] + delta[0], cell[1] + delta[1])


def is_upstream_cell(flowdir, this_cell, direction):
    """Does the neighbour in ``direction`` flow into ``this_cell``?"""
    to_cell = cell_in_direction(this_cell, direction)
    if not cell_in_raster(flowdir.shape, to_cell):
        return False
    nbr = flowdir[to_cell[0], to_cell[1]]
    if nbr == FLOWDIR_NODIR:
        return False
    return (direction + 4) % 8 == nbr


def upstream_cells(flowdir, cell):
    """Neighbours draining directly into ``cell``."""
    return [cell_in_direction(cell, d) for d in range(8) if is_upstream_cell(flowdir, cell, d)]


def trace_downstream(flowdir, cell):
    """Yield the cells on the flow path starting at ``cell`` until it leaves the raster or stops."""
    cell = tuple(cell)
    while cell and cell_in_raster(flowdir.shape, cell):
        yield cell
        delta = direction_to_delta(flowdir[cell[0], cell[1]])
        cell = (cell[0] + delta[0], cell[1] + delta[1]) if delta else None


def trace_accumulated_flow(flowdir, accum, cell):
    """Walk downstream from ``cell`` writing ``1 + sum(accum of the upstream neighbours)`` into ``accum`` and stop at
    the first cell that still has an unresolved (``<= 0``) upstream neighbour, or off the raster (flow.py:304-341).

    Per-cell host helper kept for API parity (the reference does not rebind it either at whole-stage granularity);
    the raster-wide ``accumulated_flow`` is the HIP kernel.
    """
    cell = (int(cell[0]), int(cell[1]))
    while cell_in_raster(flowdir.shape, cell):
        total = 0
        for up in upstream_cells(flowdir, cell):
            value = accum[up[0], up[1]]
            if value <= 0:
                return
            total += value
        accum[cell[0], cell[1]] = total + 1
        delta = direction_to_delta(flowdir[cell[0], cell[1]])
        if delta is None:   # a cell without direction ends the walk (the reference fails on it, flow.py:340-341)
            return
        cell = (cell[0] + delta[0], cell[1] + delta[1])


def assign_watersheds_upstream(flowdir, labelled, cell, unassigned):
    """Give every ``unassigned`` cell upstream of ``cell`` the first label met on its way down, in place
    (flow.py:367-395): depth-first over the upstream tree, carrying the label of the cell just downstream.

    Per-cell host helper kept for API parity; ``watersheds_from_labels`` (all edge cells at once) is the HIP kernel.
    """
    stack = [(int(cell[0]), int(cell[1]), unassigned)]
    while stack:
        r, c, below = stack.pop()
        lbl = labelled[r, c]
        if lbl == unassigned:
            labelled[r, c] = lbl = below
        stack.extend((u[0], u[1], lbl) for u in upstream_cells(flowdir, (r, c)))
