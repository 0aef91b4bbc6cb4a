"""The ``complete`` sequence -- mirror of ``malstroem.scripts.complete.process_all`` (reference scripts/complete.py:37-117)
as a plain function: DEM -> filled / depths / flow directions (/ accumulation) -> bluespots (filtered) -> watersheds ->
pour points -> stream network -> rain events.

The reference chains its four tools through files (each tool re-reads what the previous one wrote,
scripts/complete.py:70-117).  Here ONE device pipeline stays resident from the DEM upload to the stream walk -- or one
``BandPipeline`` per rank when the DEM is cut into row bands (``comm`` given) -- and files are written for the user only.
Outputs carry the reference's names: ``filled.tif``, ``flowdir.tif``, ``bs_depths.tif``, ``accum.tif`` (with ``accum``),
``bluespots.tif``, ``watersheds.tif`` and the vector layers ``pourpoints``, ``nodes``, ``streams``, ``events`` under
``<outdir>/vector`` (GeoJSON here; OGR formats need the reference's ``io.VectorWriter``).
"""
import ast
import logging
import os

from . import io
from .bluespots import BluespotTool
from .dem import DemTool
from .rain import RainTool
from .streams import StreamTool

logger = logging.getLogger(__name__)

_FILTER_NAMES = {"area": "area", "maxdepth": "max", "volume": "volume"}
_FILTER_NODES = (ast.Expression, ast.BoolOp, ast.And, ast.Or, ast.Compare, ast.Name, ast.Load, ast.Constant, ast.Gt, ast.GtE,
                 ast.Lt, ast.LtE, ast.Eq, ast.NotEq, ast.UnaryOp, ast.USub, ast.UAdd)


def parse_filter(filter):
    """``"area > 20.5 and (maxdepth > 0.05 or volume > 2.5)"`` -> ``f(stats) -> bool`` (reference scripts/_utils.py:16-43).

    Same vocabulary as the reference's check (the words area / maxdepth / volume / and / or, comparison operators, numbers and
    parentheses); anything else raises like there.  The expression is validated on its syntax tree instead of by character
    stripping, then compiled once."""
    if not filter:
        return lambda stats: True
    try:
        tree = ast.parse(filter.strip(), mode="eval")
    except SyntaxError:
        raise Exception('Unsupported filter statement. Illegal parts: {}'.format(filter))
    for node in ast.walk(tree):
        bad = not isinstance(node, _FILTER_NODES)
        bad = bad or (isinstance(node, ast.Name) and node.id not in _FILTER_NAMES)
        bad = bad or (isinstance(node, ast.Constant) and (isinstance(node.value, bool) or not isinstance(node.value, (int, float))))
        if bad:
            raise Exception('Unsupported filter statement. Illegal parts: {}'.format(ast.dump(node)[:60]))
    code = compile(tree, "<bluespot filter>", "eval")

    def filter_function(stats):
        return eval(code, {"__builtins__": {}}, {name: stats[key] for name, key in _FILTER_NAMES.items()})
    return filter_function


def process_all(dem, outdir, rain, accum=False, filter=None, vector=False, device=0, nodatasubst=-999):
    """Quick option to run all processes (scripts/complete.py:37-117) on one MI355X.

    ``dem``: path of the DEM GeoTIFF (metres, square cells); ``outdir``: an existing empty directory; ``rain``: rain incidents
    in mm; ``accum``: also compute the accumulated flow (pour points then sit at its maximum, bluespots.py:195-200);
    ``filter``: bluespot filter expression.  Returns a dict with the paths written and the counts the reference logs."""
    if vector:
        raise NotImplementedError("vectorisation of bluespots / watersheds (GDAL polygonize) is outside malstroem_amd's hot path")
    if not os.path.isdir(outdir) or os.listdir(outdir):
        raise ValueError("outdir isn't an empty directory")
    outvector = os.path.join(outdir, 'vector')
    filter_function = parse_filter(filter)
    dem_reader = io.RasterReader(dem, nodatasubst=nodatasubst)
    tr, crs = dem_reader.transform, dem_reader.crs
    logger.info('Processing')
    logger.info('   dem: {}'.format(dem))
    logger.info('   outdir: {}'.format(outdir))
    logger.info('   rain: {}'.format(', '.join(['{}mm'.format(r) for r in rain])))
    logger.info('   accum: {}'.format(accum))
    logger.info('   filter: {}'.format(filter))

    # Process DEM (the rasters stay on the device for the next two tools)
    filled_writer = io.RasterWriter(os.path.join(outdir, 'filled.tif'), tr, crs, nodatasubst)
    flowdir_writer = io.RasterWriter(os.path.join(outdir, 'flowdir.tif'), tr, crs)
    depths_writer = io.RasterWriter(os.path.join(outdir, 'bs_depths.tif'), tr, crs)
    accum_writer = io.RasterWriter(os.path.join(outdir, 'accum.tif'), tr, crs) if accum else None
    pipe = DemTool(dem_reader, filled_writer, flowdir_writer, depths_writer, accum_writer, device=device).process(keep_pipeline=True)
    try:
        # Process bluespots
        pourpoint_writer = io.VectorWriter('GeoJSON', outvector, 'pourpoints', None, None, crs)
        watershed_writer = io.RasterWriter(os.path.join(outdir, 'watersheds.tif'), tr, crs, 0)
        labeled_writer = io.RasterWriter(os.path.join(outdir, 'bluespots.tif'), tr, crs, 0)
        BluespotTool(input_depths=dem_reader, input_flowdir=dem_reader, input_bluespot_filter_function=filter_function,
                     input_accum=None, input_dem=dem_reader, output_labeled_raster=labeled_writer, output_pourpoints=pourpoint_writer,
                     output_watersheds_raster=watershed_writer, pipeline=pipe, device=device).process()
        nlabels = pipe.get_int("nlabels")
        # Process pourpoints: the walk runs on the resident flow directions and labels
        pourpoints_reader = io.VectorReader(outvector, pourpoint_writer.layername)
        nodes_writer = io.VectorWriter('GeoJSON', outvector, 'nodes', None, None, crs)
        streams_writer = io.VectorWriter('GeoJSON', outvector, 'streams', None, None, crs)
        StreamTool(pourpoints_reader, dem_reader, dem_reader, nodes_writer, streams_writer, pipeline=pipe).process()
    finally:
        pipe.close()
        dem_reader.close()
    # Process rain events
    nodes_reader = io.VectorReader(outvector, nodes_writer.layername)
    events_writer = io.VectorWriter('GeoJSON', outvector, 'events', None, None, crs)
    RainTool(nodes_reader, events_writer, rain).process()
    return dict(outdir=outdir, vector=outvector, nlabels=nlabels, events=events_writer.filepath,
                nodes=nodes_writer.filepath, streams=streams_writer.filepath, pourpoints=pourpoint_writer.filepath)
