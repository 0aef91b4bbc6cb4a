"""BluespotTool -- mirror of ``malstroem.bluespots`` (reference bluespots.py:23-216) on the device pipeline.

``filterbluespots`` and ``assemble_pourpoints`` keep the reference's record layout (pour point fields
of bluespots.py:75-82).  Vectorisation of rasters (GDAL polygonize) is outside the hot path and is not
provided here; pass the reference's vector writers if you need them and call the reference's
``vectorize_labels_file`` yourself.
"""
import logging

import numpy as np

from .algorithms import speedups
from .pipeline import HydroPipeline


def filterbluespots(filterfunction, cell_area, raw_bluespot_stats):
    """Apply ``filterfunction`` to every raw bluespot stat (bluespots.py:23-46).

    The function sees a dict with min, max, sum, count, volume (= sum * cell_area) and area
    (= count * cell_area) and returns True to keep the bluespot.
    """
    keepers = []
    for s in raw_bluespot_stats:
        d = dict(min=s['min'], max=s['max'], sum=s['sum'], count=s['count'])
        d['volume'] = s['sum'] * cell_area
        d['area'] = s['count'] * cell_area
        keepers.append(filterfunction(d))
    return keepers


def transform_cell_to_world(cell, transform):
    """Centre of cell (row, col) in world coordinates for a GDAL geotransform (reference vector.py:21-39)."""
    row, col = cell[0] + 0.5, cell[1] + 0.5
    x = transform[0] + col * transform[1] + row * transform[2]
    y = transform[3] + col * transform[4] + row * transform[5]
    return x, y


POURPOINT_DTYPE = np.dtype([("bspot_id", "<i8"), ("cell_row", "<i8"), ("cell_col", "<i8"), ("bspot_dmax", "<f8"), ("bspot_area", "<f8"),
                            ("bspot_vol", "<f8"), ("wshed_area", "<f8"), ("bspot_fumm", "<f8")])


def pourpoint_records(transform, pp_pix, bluespot_stats, watershed_stats, first_id=0):
    """The attributes of bluespots.py:75-82 for a run of labels as ONE structured array (``POURPOINT_DTYPE``) -- what the row-band
    path moves between ranks (65 M labels as dicts do not travel) -- computed in the arithmetic of the reference's per-label code:
    area = count * A, vol = sum * A, wshed_area = watershed cells * A, fumm = float64(1000 * vol) / float64(wshed_area)."""
    cell_area = abs(transform[1]) * abs(transform[5])
    pp_pix, bluespot_stats = np.asarray(pp_pix), np.asarray(bluespot_stats)
    n = len(pp_pix)
    rec = np.zeros(n, POURPOINT_DTYPE)
    rec["bspot_id"] = np.arange(int(first_id), int(first_id) + n, dtype=np.int64)
    rec["cell_row"], rec["cell_col"] = pp_pix["row"], pp_pix["col"]
    rec["bspot_dmax"] = bluespot_stats["max"]
    rec["bspot_area"] = bluespot_stats["count"] * cell_area
    rec["bspot_vol"] = bluespot_stats["sum"] * cell_area
    rec["wshed_area"] = np.asarray(watershed_stats) * cell_area
    with np.errstate(divide="ignore", invalid="ignore"):
        rec["bspot_fumm"] = np.float64(1000) * rec["bspot_vol"] / rec["wshed_area"]
    return rec


def pourpoint_features(records, transform):
    """GeoJSON-like pour point features (bluespots.py:49-88) of ``pourpoint_records``, one at a time."""
    for r in records:
        ix = int(r["bspot_id"])
        p = dict(bspot_id=ix, type="Feature", cell_row=int(r["cell_row"]), cell_col=int(r["cell_col"]), bspot_dmax=float(r["bspot_dmax"]),
                 bspot_area=float(r["bspot_area"]), bspot_vol=float(r["bspot_vol"]), wshed_area=float(r["wshed_area"]), bspot_fumm=float(r["bspot_fumm"]))
        coord = transform_cell_to_world((p['cell_row'], p['cell_col']), transform)
        yield dict(id=ix, geometry=dict(type='Point', coordinates=list(coord)), properties=p)


def assemble_pourpoints(transform, pp_pix, bluespot_stats, watershed_stats, first_id=0):
    """GeoJSON-like pour point features, one per label incl. background 0 (bluespots.py:49-88).  ``first_id``: the label of the
    first record (a row band assembles the features of the labels it numbered)."""
    cell_area = abs(transform[1]) * abs(transform[5])
    pour_points = []
    for ix, (pix, bstat, wcount) in enumerate(zip(pp_pix, bluespot_stats, watershed_stats), int(first_id)):
        p = dict(bspot_id=ix, type="Feature")
        p['cell_row'] = int(pix['row'])
        p['cell_col'] = int(pix['col'])
        p['bspot_dmax'] = float(bstat['max'])
        p['bspot_area'] = float(bstat['count'] * cell_area)
        p['bspot_vol'] = float(bstat['sum'] * cell_area)
        p['wshed_area'] = float(wcount * cell_area)
        p['bspot_fumm'] = float(np.float64(1000 * p['bspot_vol']) / np.float64(p['wshed_area']))
        coord = transform_cell_to_world((int(pix['row']), int(pix['col'])), transform)
        pour_points.append(dict(id=ix, geometry=dict(type='Point', coordinates=list(coord)), properties=p))
    return pour_points


class BluespotTool(object):
    """Bluespots, their local watersheds and pour points (bluespots.py:91-216).

    Same inputs/outputs as the reference.  ``pipeline``: an optional ``HydroPipeline`` left behind by
    ``DemTool.process(keep_pipeline=True)``; then nothing is re-read or recomputed (in particular the
    no-flats surface the reference computes a second time, bluespots.py:203-204, is reused).
    """

    def __init__(self, input_depths, input_flowdir, input_bluespot_filter_function,
                 output_labeled_raster, output_pourpoints, output_watersheds_raster,
                 input_accum=None, input_dem=None, output_labeled_vector=None, output_watersheds_vector=None,
                 pipeline=None, device=0):
        self.input_depths = input_depths
        self.input_flowdir = input_flowdir
        self.input_bluespot_filter_function = input_bluespot_filter_function
        self.input_accum = input_accum
        self.input_dem = input_dem
        self.output_labeled_raster = output_labeled_raster
        self.output_labeled_vector = output_labeled_vector
        self.output_pourpoints = output_pourpoints
        self.output_watersheds_raster = output_watersheds_raster
        self.output_watersheds_vector = output_watersheds_vector
        self.pipeline = pipeline
        self.device = device
        assert self.input_accum or self.input_dem or pipeline, "Either input_dem or input_accum must be specified"
        if output_labeled_vector or output_watersheds_vector:
            raise NotImplementedError("vectorisation (GDAL polygonize) is outside malstroem_amd's hot path")
        self.logger = logging.getLogger(__name__)

    def process(self):
        transform = self.input_depths.transform
        cell_width, cell_height = abs(transform[1]), abs(transform[5])
        cell_area = cell_width * cell_height
        assert abs(cell_width - cell_height) < 0.01 * abs(cell_width), "Input cells must be square"
        if not speedups.enabled:
            raise RuntimeError("malstroem_amd: HIP backend not available and there is no CPU fallback")

        pipe, own = self.pipeline, False
        if pipe is None:
            depths = self.input_depths.read()
            pipe, own = HydroPipeline(depths.shape, device=self.device), True
            if not self.input_accum:
                # pour points at the minimum of the no-flats surface (bluespots.py:203-205).  The DEM goes up FIRST: a new
                # DEM invalidates every raster derived from the previous one
                pipe.upload("dem", self.input_dem.read())
                pipe.run("noflat")
            pipe.upload("depths", depths)
            pipe.upload("flowdir", self.input_flowdir.read())
            if self.input_accum:
                pipe.upload("accum", self.input_accum.read())
        try:
            self.logger.info("Calculating unfiltered bluespots")
            pipe.run("label")
            raw_stats = pipe.raw_stats()
            self.logger.info("Number of bluespots found before filtering: {}".format(len(raw_stats) - 1))
            self.logger.info("Calculating filtered bluespots")
            keepers = filterbluespots(self.input_bluespot_filter_function, cell_area, raw_stats)
            nlabels = pipe.apply_keep(keepers)
            bluespot_stats = pipe.stats()
            self.logger.info("Number of bluespots left after filtering: {}".format(nlabels))
            self.output_labeled_raster.write(pipe.download("labels"))
            self.logger.info("Calculating watersheds and pour points")
            pipe.run("watershed", "pourpoints")
            watershed_stats = pipe.watershed_counts()
            if self.output_watersheds_raster:
                self.output_watersheds_raster.write(pipe.download("watersheds"))
            pp_pix = pipe.pourpoints()
            self.logger.info("Writing {} pour points".format(len(pp_pix)))
            pour_points = assemble_pourpoints(transform, pp_pix, bluespot_stats, watershed_stats)
            self.output_pourpoints.write_geojson_features(dict(type="FeatureCollection", features=pour_points))
            self.logger.info("Done")
        finally:
            if own:
                pipe.close()
