"""Name-compatible stand-in for ``malstroem.algorithms.speedups`` (``available``, ``enabled``,
``enable``, ``disable``; reference speedups/__init__.py:29) so that code which only checks
``speedups.enabled`` to log a warning (dem.py:62, bluespots.py:154) keeps working.
The accelerated path here is HIP, see ``malstroem_amd.algorithms.hip``.
"""
from .. import hip as _hip

__all__ = ["available", "enable", "disable", "enabled"]

available = _hip.available


def enable():
    _hip.enable()


def disable():
    _hip.disable()


def __getattr__(name):
    if name == "enabled":
        return _hip.enabled
    raise AttributeError(name)
