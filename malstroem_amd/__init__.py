"""malstroem_amd -- MI355X (gfx950) native raster-hydrology core behind malstroem's own API.

Only the hot path of SDFIdk/malstroem lives here: depression fill, no-flats
fill, D8 flow direction, flow accumulation, bluespot (connected component) labelling, per-label
reductions and watershed labelling -- hand-written HIP kernels in ``libmalstroem_hip.so`` reached
through a thin ctypes C-ABI (``include/malstroem_hip.h``).  No PyTorch, no CPU fallback.

``malstroem_amd.algorithms.{fill,flow,label}`` mirror ``malstroem.algorithms.{fill,flow,label}``;
``malstroem_amd.dem.DemTool`` / ``malstroem_amd.bluespots.BluespotTool`` mirror the reference tools
with a device-resident fast path; ``malstroem_amd.algorithms.hip.enable(pkg)`` patches an installed
``malstroem`` package the way ``malstroem.algorithms.speedups.enable()`` does for Cython.
"""
__version__ = "0.1.0"
