"""ctypes binding of libmalstroem_hip.so (C-ABI declared in include/malstroem_hip.h).

This is the only place that touches the shared library.  Loading never falls back to anything
else: if the library (or a HIP device) is missing the stage functions raise ``RuntimeError``.
"""
import ctypes
import os
import subprocess
import threading
from pathlib import Path

import numpy as np

_PKG = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("MALSTROEM_HIP_LIB", _PKG / "lib" / "libmalstroem_hip.so"))  # override: dev builds
CSRC = _PKG / "csrc"

STAT_DTYPE = np.dtype([("min", "<f8"), ("max", "<f8"), ("sum", "<f8"), ("count", "<i8")])   # _label.pyx:22-24
INDEX_DTYPE = np.dtype([("value", "<f8"), ("row", "<i8"), ("col", "<i8")])                    # _label.pyx:26-28

OK, EINVAL, EHIP, ENODEV, ELIMIT, ENOTCONV, ECOMM = 0, -1, -2, -3, -4, -5, -6

STAGE_FILL, STAGE_NOFLAT, STAGE_FLOWDIR, STAGE_ACCUM = 1, 2, 4, 8
STAGE_LABEL, STAGE_WATERSHED, STAGE_POURPOINTS, STAGE_ALL = 16, 32, 64, 0x7F
R_DEM, R_FILLED, R_DEPTHS, R_NOFLAT, R_FLOWDIR, R_ACCUM, R_LABELS, R_WATERSHEDS, R_NGDIST = range(9)
RASTER_DTYPE = {R_DEM: np.float32, R_FILLED: np.float32, R_DEPTHS: np.float32, R_NOFLAT: np.float64,
                R_FLOWDIR: np.uint8, R_ACCUM: np.float64, R_LABELS: np.int32, R_WATERSHEDS: np.int32, R_NGDIST: np.uint32}

# every symbol include/malstroem_hip.h declares (checked by tests/test_cabi.py)
SYMBOLS = [
    "mhip_last_error", "mhip_version", "mhip_device_count", "mhip_set_device", "mhip_copy_bandwidth", "mhip_read_bandwidth",
    "mhip_fill_f32", "mhip_fill_noflat_f64", "mhip_short_diag", "mhip_depths_f32", "mhip_d8_f64", "mhip_accum",
    "mhip_ccl8_f32", "mhip_ccl8_u8", "mhip_relabel_keep", "mhip_keep_mask", "mhip_label_stats_f32", "mhip_label_stats_f64",
    "mhip_label_argmin_f64", "mhip_label_argmax_f64", "mhip_label_count", "mhip_label_max", "mhip_watersheds_i32",
    "mhip_trace_downstream_i32", "mhip_ctx_trace_downstream", "mhip_rain_events", "mhip_band_forest_solve", "mhip_band_ws_resolve",
    "mhip_ctx_create", "mhip_comm_unique_id", "mhip_comm_available", "mhip_ctx_create_band", "mhip_ctx_destroy", "mhip_ctx_upload_dem",
    "mhip_ctx_upload", "mhip_ctx_download", "mhip_ctx_upload_rows", "mhip_ctx_download_rows", "mhip_ctx_run", "mhip_ctx_sync", "mhip_ctx_stage_ms",
    "mhip_ctx_kernel_ms", "mhip_ctx_get_i64", "mhip_ctx_get_f64", "mhip_ctx_raw_stats", "mhip_ctx_apply_keep",
    "mhip_ctx_stats", "mhip_ctx_watershed_counts", "mhip_ctx_pourpoints",
    "mhip_ctx_band_info", "mhip_ctx_get_edge_row", "mhip_ctx_set_halo_row", "mhip_ctx_get_edge_row_dev",
    "mhip_ctx_set_halo_row_dev", "mhip_ctx_get_edge_rows", "mhip_ctx_set_halo_rows", "mhip_ctx_dem_minmax",
    "mhip_ctx_fill_begin", "mhip_ctx_fill_batch", "mhip_ctx_fill_halo_changed", "mhip_ctx_fill_certify", "mhip_ctx_fill_end",
    "mhip_ctx_geo_begin", "mhip_ctx_geo_batch", "mhip_ctx_geo_halo_changed", "mhip_ctx_geo_end", "mhip_ctx_fill_attach", "mhip_ctx_noflat_verify",
    "mhip_ctx_zero_raster", "mhip_ctx_band_accum_boundary", "mhip_ctx_band_ccl_local", "mhip_ctx_band_ccl_begin", "mhip_ctx_band_ccl_finish", "mhip_ctx_band_relabel", "mhip_ctx_band_relabel_sparse", "mhip_ctx_band_relabel_range", "mhip_ctx_band_trace", "mhip_ctx_band_watershed_local",
    "mhip_ctx_band_apply_neg_lut", "mhip_ctx_band_records", "mhip_ctx_band_fetch", "mhip_ctx_band_gather",
    "mhip_ctx_band_foreign_counts", "mhip_ctx_side_begin", "mhip_ctx_side_end",
    "mhip_ctx_has_comm", "mhip_ctx_exchange_halo", "mhip_ctx_exchange_edge_rows", "mhip_ctx_comm_add_side", "mhip_band_union_find", "mhip_band_accum_pairs", "mhip_band_accum_solve", "mhip_band_label_pairs",
    "mhip_band_label_merge", "mhip_band_ws_publish", "mhip_band_ws_lut", "mhip_band_merge_records", "mhip_tg_create", "mhip_tg_destroy", "mhip_tg_barrier", "mhip_tg_allreduce_max", "mhip_tg_offer", "mhip_tg_take", "mhip_shm_barrier", "mhip_ctx_allreduce_max",
]

_lib = None
# ctypes releases the GIL.  The library is thread safe (thread-local error text, locked buffer pool); a context is driven by
# one thread at a time, except for the side-stream bracket mhip_ctx_side_begin / _end.


def build(force=False):
    """Compile libmalstroem_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.hpp")) + [_PKG.parent / "include" / "malstroem_hip.h"]
    if not force and LIB_PATH.exists() and all(LIB_PATH.stat().st_mtime >= s.stat().st_mtime for s in srcs):
        return LIB_PATH
    subprocess.check_call(["make", "-C", str(CSRC), "-j8"] + (["-B"] if force else []))
    return LIB_PATH


def load():
    """Load the shared library (no device needed for loading)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise RuntimeError("libmalstroem_hip.so is not built (%s); run malstroem_amd._lib.build() -- "
                               "there is no CPU fallback" % LIB_PATH)
        lib = ctypes.CDLL(str(LIB_PATH))
        lib.mhip_last_error.restype = ctypes.c_char_p
        lib.mhip_version.restype = ctypes.c_char_p
        _lib = lib
    return _lib


def device_count():
    try:
        return int(load().mhip_device_count())
    except (OSError, RuntimeError):
        return 0


def check(rc, what):
    if rc == OK:
        return
    msg = load().mhip_last_error().decode("utf-8", "replace")
    text = "%s: %s" % (what, msg)
    if rc == EINVAL:
        raise ValueError(text)
    if rc == ELIMIT:
        raise OverflowError(text)
    raise RuntimeError(text)


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def i64(v):
    return ctypes.c_int64(int(v))


def call(name, *args):
    rc = getattr(load(), name)(*args)
    check(rc, name)
