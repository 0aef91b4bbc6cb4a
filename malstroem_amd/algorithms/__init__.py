"""Mirror of ``malstroem.algorithms``: fill, flow, label (+ dtypes, _raster_utils, hip, speedups).

Importing the package enables the HIP backend when a device is present, like the reference enables
its Cython speedups at import (speedups/__init__.py:103-104).
"""
from . import dtypes, _raster_utils, fill, flow, label, net  # noqa: F401
from . import hip, speedups  # noqa: F401
