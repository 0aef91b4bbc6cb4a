"""Small raster index helpers (reference malstroem/algorithms/_raster_utils.py:18-60)."""


def cell_in_raster(shape, cell):
    """True if (row, col) lies inside a raster of ``shape`` (rows, cols)."""
    return 0 <= cell[0] < shape[0] and 0 <= cell[1] < shape[1]


def edge_cell_indexes(shape):
    """Yield the edge cells: (0,c),(maxr,c) per column, then (r,0),(r,maxc) per inner row
    (the reference's iteration order, _raster_utils.py:55-60)."""
    maxr, maxc = shape[0] - 1, shape[1] - 1
    for c in range(maxc + 1):
        yield (0, c)
        yield (maxr, c)
    for r in range(1, maxr):
        yield (r, 0)
        yield (r, maxc)
