"""dtype contract of the raster hot path (reference malstroem/algorithms/dtypes.py:20-31)."""
import numpy as np

DTYPE_DTM = np.float32          # input DEM
DTYPE_FILL = DTYPE_DTM          # filled DEM
DTYPE_FILLNOFLAT = np.float64   # no-flats surface
DTYPE_FLOWDIR = np.uint8        # AGNPS flow direction codes
DTYPE_ACCUM = np.float64        # accumulated flow
DTYPE_LABEL = np.int32          # scipy.ndimage.label output dtype (reference tests/test_raster_label.py:11)
