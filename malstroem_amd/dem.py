"""DemTool -- mirror of ``malstroem.dem.DemTool`` (reference dem.py:20-93) on the device pipeline.

Same constructor, same ``process()`` contract (reads the DEM from a raster reader, writes filled DEM,
bluespot depths, flow directions and optionally accumulated flow to raster writers).  All rasters are
computed in one device-resident pass; nothing is recomputed and nothing goes through disk in between.
"""
import logging

from .algorithms import dtypes, speedups
from .pipeline import HydroPipeline


class DemTool(object):
    """Calculate filled DEM, flow directions, bluespot depths and optionally accumulated flow.

    Parameters mirror the reference: ``input_dem`` needs ``read()`` and ``transform``; the outputs need
    ``write(array)``.  Input x, y, z must be in meters and cells must be square.
    """

    def __init__(self, input_dem, output_filled, output_flowdir, output_depths, output_accum=None, device=0):
        self.input_dem = input_dem
        self.output_filled = output_filled
        self.output_flowdir = output_flowdir
        self.output_depths = output_depths
        self.output_accum = output_accum
        self.device = device
        self.logger = logging.getLogger(__name__)

    def process(self, keep_pipeline=False):
        streaming = hasattr(self.input_dem, "iter_windows") and hasattr(self.input_dem, "shape")
        dem = None if streaming else self.input_dem.read().astype(dtypes.DTYPE_DTM, casting='same_kind', copy=False)
        transform = self.input_dem.transform
        assert abs(abs(transform[1]) - abs(transform[5])) < 0.01 * abs(transform[1]), "Input cells must be square"
        if not speedups.enabled:
            raise RuntimeError("malstroem_amd: HIP backend not available and there is no CPU fallback")

        pipe = HydroPipeline(self.input_dem.shape if streaming else dem.shape, device=self.device)
        try:
            if streaming:
                # windowed reader (malstroem_amd.io.RasterReader): the DEM goes to the device window by window, the results
                # come back the same way -- the host never holds a whole raster (reference io.py:60-73 / 141-159 do)
                for row0, window in self.input_dem.iter_windows():
                    pipe.upload_rows("dem", row0, window.astype(dtypes.DTYPE_DTM, casting='same_kind', copy=False))
            else:
                pipe.upload("dem", dem)
            self.logger.info("Calculating filled DEM and bluespot depths")
            pipe.run("fill")
            pipe.download_to("filled", self.output_filled)
            pipe.download_to("depths", self.output_depths)
            self.logger.info("Calculating flow directions")
            pipe.run("noflat", "flowdir")
            pipe.download_to("flowdir", self.output_flowdir)
            if self.output_accum:
                self.logger.info("Calculating flow accumulation")
                pipe.run("accum")
                pipe.download_to("accum", self.output_accum)
            self.logger.info("Done")
        except Exception:
            pipe.close()
            raise
        if keep_pipeline:
            return pipe   # BluespotTool can continue on the resident rasters
        pipe.close()
        return None
