"""CPU-only: the C-ABI shared library loads and exports every symbol include/malstroem_hip.h declares;
without a GPU every compute entry point fails loudly (no CPU fallback)."""
import ctypes
import os
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def lib():
    from malstroem_amd import _lib
    _lib.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    from malstroem_amd import _lib
    header = (ROOT / "include" / "malstroem_hip.h").read_text()
    declared = set(re.findall(r"\b(mhip_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in sorted(declared):
        assert hasattr(lib, name), name


def test_record_layouts_match_numpy():
    from malstroem_amd import _lib
    assert _lib.STAT_DTYPE.itemsize == 32 and _lib.INDEX_DTYPE.itemsize == 24   # _label.pyx:22-28 packed structs


def test_version_and_error_strings(lib):
    assert b"gfx950" in lib.mhip_version()
    assert isinstance(lib.mhip_last_error(), bytes)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_cpu_fallback_without_gpu(lib):
    import malstroem_amd.algorithms as alg
    assert lib.mhip_device_count() == 0
    assert not alg.hip.available and not alg.speedups.enabled
    dem = np.zeros((8, 8), dtype=np.float32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        alg.fill.fill_terrain(dem)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        alg.flow.accumulated_flow(np.zeros((8, 8), dtype=np.uint8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        alg.label.connected_components(dem)


def test_argument_errors_mirror_reference_dtype_checks():
    import malstroem_amd.algorithms as alg
    with pytest.raises(ValueError, match="dtype mismatch"):
        alg.fill.fill_terrain(np.zeros((8, 8), dtype=np.float64))       # _fill.pyx:30 takes float32 only
    with pytest.raises(ValueError, match="dtype mismatch"):
        alg.flow.terrain_flowdirection(np.zeros((8, 8), dtype=np.float32))  # _flow.pyx:99 takes float64 only
    with pytest.raises(ValueError, match="dtype mismatch"):
        alg.flow.accumulated_flow(np.zeros((8, 8), dtype=np.int32))     # _flow.pyx:257 takes uint8 only


def test_host_helpers_match_reference_semantics():
    # reference flow.py:170-301 / _raster_utils.py:18-60 helpers are plain host code
    from malstroem_amd.algorithms import _raster_utils, flow
    fd = np.full((4, 5), flow.FLOWDIR_NODIR, dtype=np.uint8)
    fd[1, 1] = flow.FLOWDIR_RIGHT
    fd[1, 2] = flow.FLOWDIR_DOWN_RIGHT
    fd[2, 3] = flow.FLOWDIR_DOWN
    assert list(flow.trace_downstream(fd, (1, 1))) == [(1, 1), (1, 2), (2, 3), (3, 3)]
    assert flow.upstream_cells(fd, (1, 2)) == [(1, 1)]
    assert flow.direction_to_delta(flow.FLOWDIR_NODIR) is None and flow.direction_to_delta(7) == (-1, -1)
    assert list(_raster_utils.edge_cell_indexes((3, 3))) == [(0, 0), (2, 0), (0, 1), (2, 1), (0, 2), (2, 2), (1, 0), (1, 2)]
    assert _raster_utils.cell_in_raster((3, 3), (2, 2)) and not _raster_utils.cell_in_raster((3, 3), (3, 0))
    f2 = np.full((3, 3), 8, dtype=np.uint8)
    flow.set_edges_flow_outward(f2)
    assert f2.tolist() == [[7, 0, 1], [6, 8, 2], [5, 4, 3]]


REFERENCE = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REFERENCE + "/malstroem/algorithms"), reason="reference checkout not on this box")
def test_enable_patches_the_real_reference_package(monkeypatch):
    """hip.enable(target=malstroem.algorithms) on the reference's own (pure-Python) package: the 13 whole-stage functions
    are rebound like speedups.enable() rebinds its 8 (speedups/__init__.py:37-77), with identical call signatures, and
    disable() restores them.  Runs without a GPU (nothing is called); `available` is forced for the duration."""
    import inspect
    import sys
    import types
    monkeypatch.setattr(sys, "dont_write_bytecode", True)       # never write into the read-only reference tree
    monkeypatch.syspath_prepend(REFERENCE)
    import malstroem.algorithms as ref_alg
    from malstroem.algorithms import fill as rfill, flow as rflow, label as rlabel  # noqa: F401
    import malstroem_amd.algorithms as alg
    hip = alg.hip
    originals = {(m, a): getattr(getattr(ref_alg, m), a) for (m, a) in hip._PATCH}
    assert len(originals) == 13
    was_enabled = hip.enabled
    monkeypatch.setattr(hip, "available", True)
    hip.disable()
    other = types.SimpleNamespace(fill=types.SimpleNamespace(**{a: None for (m, a) in hip._PATCH if m == "fill"}),
                                  flow=types.SimpleNamespace(**{a: None for (m, a) in hip._PATCH if m == "flow"}),
                                  label=types.SimpleNamespace(**{a: None for (m, a) in hip._PATCH if m == "label"}))
    try:
        hip.enable(target=ref_alg)
        for (m, a), fn in hip._PATCH.items():
            bound = getattr(getattr(ref_alg, m), a)
            assert bound is fn and bound is not originals[(m, a)], (m, a)
            ours = list(inspect.signature(fn).parameters.values())
            theirs = list(inspect.signature(originals[(m, a)]).parameters.values())
            # same names, order, kinds and defaults; ours may only ADD trailing keyword arguments with defaults
            assert [(p.name, p.kind, p.default) for p in ours[:len(theirs)]] == [(p.name, p.kind, p.default) for p in theirs], (m, a)
            assert all(p.default is not inspect.Parameter.empty for p in ours[len(theirs):]), (m, a)
        # functions the reference keeps un-patched stay its own, and exist in the mirror with the same signatures
        for name in ("trace_accumulated_flow", "assign_watersheds_upstream", "upstream_cells", "trace_downstream"):
            assert getattr(ref_alg.flow, name).__module__ == "malstroem.algorithms.flow"
            assert (list(inspect.signature(getattr(alg.flow, name)).parameters) ==
                    list(inspect.signature(getattr(ref_alg.flow, name)).parameters)), name
        hip.enable(target=ref_alg)                       # idempotent (speedups/__init__.py:44-45)
        assert len(hip._orig) == 13
        hip.enable(target=other)                         # a second target is patched too, not silently ignored
        assert other.fill.fill_terrain is alg.fill.fill_terrain and len(hip._orig) == 26
        with pytest.raises(AttributeError):
            hip.enable(target=types.SimpleNamespace(fill=None, flow=None, label=None))
    finally:
        hip.disable()
    for (m, a), fn in originals.items():
        assert getattr(getattr(ref_alg, m), a) is fn, (m, a)
    assert other.fill.fill_terrain is None
    monkeypatch.undo()
    if was_enabled:
        hip.enable()
