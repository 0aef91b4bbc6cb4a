"""Device-resident pipeline: one upload, every stage in HBM, downloads only for the writers.

Python face of ``mhip_ctx`` (include/malstroem_hip.h).  It runs the DemTool sequence
(reference dem.py:53-93) and the BluespotTool sequence (bluespots.py:138-216) without the GeoTIFF
round trips the reference makes between its tools (scripts/complete.py:70-81) and without running the
no-flats fill twice (bluespots.py:203-204).
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import (INDEX_DTYPE, R_ACCUM, R_DEM, R_DEPTHS, R_FILLED, R_FLOWDIR, R_LABELS, R_NGDIST, R_NOFLAT, R_WATERSHEDS,
                   RASTER_DTYPE, STAGE_ACCUM, STAGE_FILL, STAGE_FLOWDIR, STAGE_LABEL, STAGE_NOFLAT,
                   STAGE_POURPOINTS, STAGE_WATERSHED, STAT_DTYPE)

STAGES = {"fill": STAGE_FILL, "noflat": STAGE_NOFLAT, "flowdir": STAGE_FLOWDIR, "accum": STAGE_ACCUM,
          "label": STAGE_LABEL, "watershed": STAGE_WATERSHED, "pourpoints": STAGE_POURPOINTS}
RASTERS = {"dem": R_DEM, "filled": R_FILLED, "depths": R_DEPTHS, "noflat": R_NOFLAT, "flowdir": R_FLOWDIR,
           "accum": R_ACCUM, "labels": R_LABELS, "watersheds": R_WATERSHEDS, "ngdist": R_NGDIST}


class HydroPipeline(object):
    """Rasters of one H x W DEM (or one row band of it) resident on one MI355X."""

    def __init__(self, shape, device=0):
        self.shape = (int(shape[0]), int(shape[1]))
        self._ctx = ctypes.c_void_p()
        _lib.call("mhip_ctx_create", ctypes.byref(self._ctx), _lib.i64(self.shape[0]), _lib.i64(self.shape[1]), int(device))

    def close(self):
        if self._ctx:
            _lib.call("mhip_ctx_destroy", self._ctx)
            self._ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data movement ---------------------------------------------------------------------------
    def upload(self, name, array):
        which = RASTERS[name]
        a = np.ascontiguousarray(array, dtype=RASTER_DTYPE[which])
        if a.shape != self.shape:
            raise ValueError("raster shape %s does not match the pipeline shape %s" % (a.shape, self.shape))
        _lib.call("mhip_ctx_upload", self._ctx, which, _lib.ptr(a))

    def download(self, name):
        which = RASTERS[name]
        out = np.empty(self.shape, dtype=RASTER_DTYPE[which])
        _lib.call("mhip_ctx_download", self._ctx, which, _lib.ptr(out))
        return out

    # ---- windowed data movement (malstroem_amd.io readers / writers): one window on the host, whatever the raster's size
    def upload_rows(self, name, row0, array):
        which = RASTERS[name]
        a = np.ascontiguousarray(array, dtype=RASTER_DTYPE[which])
        if a.ndim != 2 or a.shape[1] != self.shape[1]:
            raise ValueError("window must be full-width rows of the pipeline's raster")
        _lib.call("mhip_ctx_upload_rows", self._ctx, which, _lib.i64(row0), _lib.i64(a.shape[0]), _lib.ptr(a))

    def download_rows(self, name, row0, nrows):
        which = RASTERS[name]
        out = np.empty((int(nrows), self.shape[1]), dtype=RASTER_DTYPE[which])
        _lib.call("mhip_ctx_download_rows", self._ctx, which, _lib.i64(row0), _lib.i64(nrows), _lib.ptr(out))
        return out

    def upload_from(self, name, reader, max_rows=None):
        """Stream raster ``name`` from a reader with ``iter_windows`` (malstroem_amd.io.RasterReader); falls back to ``read()``."""
        if hasattr(reader, "iter_windows"):
            for row0, window in reader.iter_windows(max_rows):
                self.upload_rows(name, row0, window)
        else:
            self.upload(name, reader.read())

    def download_to(self, name, writer, max_rows=4096):
        """Stream raster ``name`` into a writer with ``open`` / ``write_window`` / ``close``; falls back to ``write(array)``."""
        if hasattr(writer, "write_window"):
            writer.open(self.shape, RASTER_DTYPE[RASTERS[name]])
            for row0 in range(0, self.shape[0], int(max_rows)):
                n = min(int(max_rows), self.shape[0] - row0)
                writer.write_window(row0, self.download_rows(name, row0, n))
            writer.close()
        else:
            writer.write(self.download(name))

    # ---- stages ------------------------------------------------------------------------------------
    def run(self, *stages):
        mask = 0
        for s in stages:
            mask |= STAGES[s] if isinstance(s, str) else int(s)
        _lib.call("mhip_ctx_run", self._ctx, mask)

    def sync(self):
        _lib.call("mhip_ctx_sync", self._ctx)

    def stage_ms(self, stage):
        ms = ctypes.c_float(0)
        _lib.call("mhip_ctx_stage_ms", self._ctx, STAGES[stage], ctypes.byref(ms))
        return ms.value

    def kernel_ms(self, family):
        ms, n = ctypes.c_float(0), ctypes.c_int32(0)
        _lib.call("mhip_ctx_kernel_ms", self._ctx, family.encode(), ctypes.byref(ms), ctypes.byref(n))
        return ms.value, n.value

    @staticmethod
    def copy_bandwidth(nbytes=1 << 30, reps=10):
        """GB/s (read + written) of a plain device-to-device copy: the measured ceiling the roofline fractions sit under."""
        gbs = ctypes.c_double(0)
        _lib.call("mhip_copy_bandwidth", _lib.i64(nbytes), ctypes.c_int32(reps), ctypes.byref(gbs))
        return gbs.value

    @staticmethod
    def read_bandwidth(nbytes=1 << 30, reps=10):
        """GB/s of a read-only stream of 16-byte loads (what a kernel that mostly reads, like D8, can hope for)."""
        gbs = ctypes.c_double(0)
        _lib.call("mhip_read_bandwidth", _lib.i64(nbytes), ctypes.c_int32(reps), ctypes.byref(gbs))
        return gbs.value

    def get_int(self, key):
        v = ctypes.c_int64(0)
        _lib.call("mhip_ctx_get_i64", self._ctx, key.encode(), ctypes.byref(v))
        return v.value

    def get_float(self, key):
        v = ctypes.c_double(0)
        _lib.call("mhip_ctx_get_f64", self._ctx, key.encode(), ctypes.byref(v))
        return v.value

    # ---- label bookkeeping (bluespots.py:159-172) -------------------------------------------------
    def raw_stats(self):
        rec = np.zeros(self.get_int("nlabels_raw") + 1, dtype=STAT_DTYPE)
        _lib.call("mhip_ctx_raw_stats", self._ctx, _lib.ptr(rec))
        return rec

    def apply_keep(self, keep=None):
        """``keep``: sequence of nlabels_raw+1 booleans (index 0 = background, always dropped) or None."""
        if keep is None:
            _lib.call("mhip_ctx_apply_keep", self._ctx, None)
        else:
            k = np.ascontiguousarray(np.asarray(keep).astype(bool)).view(np.uint8)
            if k.size != self.get_int("nlabels_raw") + 1:
                raise ValueError("keep must have nlabels_raw + 1 entries")
            _lib.call("mhip_ctx_apply_keep", self._ctx, _lib.ptr(k))
        return self.get_int("nlabels")

    def stats(self):
        rec = np.zeros(self.get_int("nlabels") + 1, dtype=STAT_DTYPE)
        _lib.call("mhip_ctx_stats", self._ctx, _lib.ptr(rec))
        return rec

    def watershed_counts(self):
        out = np.zeros(self.get_int("nlabels") + 1, dtype=np.int64)
        _lib.call("mhip_ctx_watershed_counts", self._ctx, _lib.ptr(out))
        return out

    def pourpoints(self):
        rec = np.zeros(self.get_int("nlabels") + 1, dtype=INDEX_DTYPE)
        _lib.call("mhip_ctx_pourpoints", self._ctx, _lib.ptr(rec))
        return rec
