"""CPU oracle for the malstroem raster hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package.  The product (``malstroem_amd``) never does.  See
``oracle/malstroem_oracle.c`` for the restated algorithms and their reference citations.
"""
from .oracle import *  # noqa: F401,F403
