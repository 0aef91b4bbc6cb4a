"""HIP backend switch -- the MI355X counterpart of ``malstroem.algorithms.speedups``
(reference speedups/__init__.py:22-104: ``available``, ``enabled``, ``enable()``, ``disable()``).

``malstroem_amd.algorithms`` is HIP-only, so inside this package ``enable()`` merely verifies that the
library and a device are usable.  Its real job is ``enable(target=malstroem.algorithms)``: rebinding the
whole-stage functions of an installed reference package to the HIP implementations (same attribute
rebinding mechanism as the Cython speedups, at the granularity a GPU needs -- SURVEY.md 8b).
"""
import warnings

from ... import _lib
from .. import fill as _fill, flow as _flow, label as _label

__all__ = ["available", "enable", "disable", "enabled"]

available = _lib.LIB_PATH.exists() and _lib.device_count() > 0
enabled = False
_orig = []   # (module object, attribute name, original function)
_targets = []   # packages patched by enable(target=...) since the last disable()

# (module name, attribute) -> HIP implementation
_PATCH = {
    ("fill", "fill_terrain"): _fill.fill_terrain,
    ("fill", "fill_terrain_no_flats"): _fill.fill_terrain_no_flats,
    ("fill", "minimum_safe_short_and_diag"): _fill.minimum_safe_short_and_diag,
    ("flow", "terrain_flowdirection"): _flow.terrain_flowdirection,
    ("flow", "_terrain_flow"): _flow._terrain_flow,
    ("flow", "accumulated_flow"): _flow.accumulated_flow,
    ("flow", "watersheds_from_labels"): _flow.watersheds_from_labels,
    ("label", "connected_components"): _label.connected_components,
    ("label", "label_stats"): _label.label_stats,
    ("label", "label_min_index"): _label.label_min_index,
    ("label", "label_max_index"): _label.label_max_index,
    ("label", "keep_labels"): _label.keep_labels,
    ("label", "label_count"): _label.label_count,
}


def enable(target=None):
    """Enable the HIP backend.  ``target``: an imported ``malstroem.algorithms`` package to patch."""
    global enabled
    if not available:
        warnings.warn("malstroem_amd HIP backend not available (library not built or no MI355X visible)",
                      RuntimeWarning)
        return
    if target is not None and not any(t is target for t in _targets):
        # idempotent per target like speedups.enable (speedups/__init__.py:44-45); a second, different target is
        # patched as well (both are restored by disable())
        missing = [(mod, attr) for (mod, attr) in _PATCH if not hasattr(getattr(target, mod, None), attr)]
        if missing:
            raise AttributeError("enable(target): %r lacks %s" % (target, ", ".join("%s.%s" % m for m in missing)))
        for (mod, attr), fn in _PATCH.items():
            m = getattr(target, mod)
            _orig.append((m, attr, getattr(m, attr)))
            setattr(m, attr, fn)
        _targets.append(target)
    enabled = True


def disable():
    """Undo ``enable(target)`` (restores the reference functions) and mark the backend disabled."""
    global enabled
    for m, attr, fn in reversed(_orig):
        setattr(m, attr, fn)
    del _orig[:]
    del _targets[:]
    enabled = False


if available:
    enable()
