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
import operator
import os

import numpy as np

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
        keep_all = lambda stats: True
        keep_all.vectorized = lambda stats: np.ones(len(stats["area"]), dtype=bool)
        return keep_all
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

    _CMP = {ast.Gt: operator.gt, ast.GtE: operator.ge, ast.Lt: operator.lt, ast.LtE: operator.le, ast.Eq: operator.eq, ast.NotEq: operator.ne}

    def vec(node, env):
        if isinstance(node, ast.Expression):
            return vec(node.body, env)
        if isinstance(node, ast.BoolOp):
            parts = [np.asarray(vec(v, env), dtype=bool) for v in node.values]
            return (np.logical_and if isinstance(node.op, ast.And) else np.logical_or).reduce(np.broadcast_arrays(*parts))
        if isinstance(node, ast.Compare):
            left, out = vec(node.left, env), True
            for op, right in zip(node.ops, node.comparators):
                right = vec(right, env)
                out = np.logical_and(out, _CMP[type(op)](left, right))
                left = right
            return out
        if isinstance(node, ast.UnaryOp):
            v = vec(node.operand, env)
            return -v if isinstance(node.op, ast.USub) else +v
        if isinstance(node, ast.Name):
            return env[node.id]
        return node.value

    def vectorized(stats):
        """the same predicate on arrays (``stats``: mapping with max / area / volume arrays): one boolean per bluespot -- what the
        row-band path evaluates on the records of millions of labels instead of calling the function once per label"""
        env = {name: np.asarray(stats[key]) for name, key in _FILTER_NAMES.items()}
        n = len(env["area"])
        return np.broadcast_to(np.asarray(vec(tree, env), dtype=bool), (n,)).copy()
    filter_function.vectorized = vectorized
    return filter_function


def process_all(dem, outdir, rain, accum=False, filter=None, vector=False, device=0, nodatasubst=-999, comm=None, backend_factory=None):
    """Quick option to run all processes (scripts/complete.py:37-117) on one MI355X -- or, with ``comm`` (a
    ``malstroem_amd.distributed.Comm`` of more than one rank; every rank calls this function), on the row bands of one DEM, one band per
    rank: see ``_process_all_bands``.

    ``dem``: path of the DEM GeoTIFF (metres, square cells); ``outdir``: an existing empty directory; ``rain``: rain incidents
    in mm; ``accum``: also compute the accumulated flow (pour points then sit at its maximum, bluespots.py:195-200);
    ``filter``: bluespot filter expression.  Returns a dict with the paths written and the counts the reference logs."""
    if vector:
        raise NotImplementedError("vectorisation of bluespots / watersheds (GDAL polygonize) is outside malstroem_amd's hot path")
    if comm is not None and comm.size > 1:
        return _process_all_bands(dem, outdir, rain, accum, filter, comm, device, nodatasubst, backend_factory)
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


def _process_all_bands(dem, outdir, rain, accum, filter, comm, device, nodatasubst, backend_factory):
    """``complete`` on row bands (BASELINE configs[4]): every rank reads its rows of the DEM file, the band chain runs on the resident
    bands (fills with halo exchange, D8, accumulation, labels merged across the seams), then -- what the reference does between its
    tools through files -- on the bands directly: the bluespot filter (every rank decides about the labels it numbered), watersheds,
    pour points (records merged across bands), the stream walk (walkers handed over at the seams) and, on rank 0, the junction
    surgery, the rain events and all writing.  Returns the same dict as ``process_all`` on rank 0, ``None`` elsewhere."""
    from .algorithms import dtypes, net
    from .bluespots import POURPOINT_DTYPE, pourpoint_features, pourpoint_records
    from .distributed import BandPipeline
    from .streams import nodes_to_features
    root = comm.rank == 0
    err = None
    if root and (not os.path.isdir(outdir) or os.listdir(outdir)):
        err = ValueError("outdir isn't an empty directory")
    if comm.allreduce_max(1.0 if err else 0.0) > 0.0:
        raise err or ValueError("outdir isn't an empty directory (rank 0)")
    outvector = os.path.join(outdir, 'vector')
    filter_function = parse_filter(filter)
    reader = io.RasterReader(dem, nodatasubst=nodatasubst)
    tr, crs = reader.transform, reader.crs
    cell_area = abs(tr[1]) * abs(tr[5])
    assert abs(abs(tr[1]) - abs(tr[5])) < 0.01 * abs(tr[1]), "Input cells must be square"
    pipe = BandPipeline(comm, reader.shape, device=device, backend_factory=backend_factory)
    try:
        pipe.upload_dem(reader.read_window(pipe.row0, pipe.nrows).astype(dtypes.DTYPE_DTM, casting='same_kind', copy=False))
        reader.close()

        def write(name, filename, nodata=None):
            # every rank writes the tile rows that start in its band into the ONE file (io.BandRasterWriter): the reference's tools
            # write the raster they hold (io.py:141-159), and here nobody holds it -- nor may a rank become the funnel for it.
            # Bands so thin that a tile row spans three of them (tests): gathered to rank 0 in row windows, as until round 3
            if io.BandRasterWriter.supports(extents, reader.shape[0]):
                pipe.write_raster(name, io.BandRasterWriter(os.path.join(outdir, filename), tr, crs, nodata))
                return
            w = io.RasterWriter(os.path.join(outdir, filename), tr, crs, nodata) if root else None
            opened = False
            for row0, rows in pipe.gather_rows(name):
                if root:
                    if not opened:
                        w.open(pipe.H_W, rows.dtype)
                        opened = True
                    w.write_window(row0, rows)
            if root:
                w.close()
        pipe.H_W = reader.shape
        extents = [tuple(e) for e in comm.allgather((int(pipe.row0), int(pipe.nrows)))]
        logger.info("Calculating filled DEM and bluespot depths")
        pipe.fill()
        write("filled", "filled.tif", nodatasubst)
        write("depths", "bs_depths.tif")
        logger.info("Calculating flow directions")
        pipe.noflat()
        pipe.flowdir()
        write("flowdir", "flowdir.tif")
        if accum:
            logger.info("Calculating flow accumulation")
            pipe.accum()
            write("accum", "accum.tif")
        logger.info("Calculating unfiltered bluespots")
        nraw = pipe.label()
        logger.info("Number of bluespots found before filtering: {}".format(nraw))

        def keep(records):       # bluespots.py:23-46 on record arrays
            stats = dict(max=records["max"], area=records["count"] * cell_area, volume=records["sum"] * cell_area)
            if hasattr(filter_function, "vectorized"):
                return filter_function.vectorized(stats)
            return np.array([bool(filter_function(dict(min=r["min"], max=r["max"], sum=r["sum"], count=r["count"], volume=r["sum"] * cell_area,
                                                       area=r["count"] * cell_area))) for r in records], dtype=bool)
        nlabels = pipe.filter(keep)
        logger.info("Number of bluespots left after filtering: {}".format(nlabels))
        write("labels", "bluespots.tif", 0)
        logger.info("Calculating watersheds and pour points")
        pipe.watershed()
        write("watersheds", "watersheds.tif", 0)
        stats, counts, pour = pipe.stats(), pipe.watershed_counts(), pipe.pourpoints(use_accum=bool(accum))
        # pour points: every rank the records of its own labels (rank 0 the background record in front: bluespots.py:49-88 emits
        # index 0 as well) as ONE structured array -- the ranks trade arrays, and only rank 0 ever turns records into features
        lo = stats["first_label"]
        recs = pourpoint_records(tr, pour["records"], stats["records"], counts["records"], first_id=lo)
        if root:
            recs = np.concatenate([pourpoint_records(tr, [pour["background"]], [stats["background"]], [counts["background"]], first_id=0), recs])
        recs = np.concatenate([np.asarray(part, dtype=POURPOINT_DTYPE) for part in comm.allgather(recs)])
        pourpoint_writer = io.VectorWriter('GeoJSON', outvector, 'pourpoints', None, None, crs)
        feats = None
        if root:
            logger.info("Writing {} pour points".format(len(recs)))
            feats = list(pourpoint_features(recs, tr))
            pourpoint_writer.write_geojson_features(dict(type="FeatureCollection", features=feats))
        # stream network: walkers start at every pour point (also the background's) and are handed over at the seams
        pix = np.stack([recs["cell_row"], recs["cell_col"]], axis=1)
        labels_next, geoms = pipe.trace_downstream(pix, 0, geometry=True)
    finally:
        pipe.close()
    if not root:
        comm.allreduce_max(0.0)       # (rank 0 is writing: leave together)
        return None
    try:
        ids = recs["bspot_id"].tolist()
        nodes = net.network_from_walks(ids, pix.tolist(), labels_next, geoms, next_label=nlabels + 1)
        nodes_writer = io.VectorWriter('GeoJSON', outvector, 'nodes', None, None, crs)
        streams_writer = io.VectorWriter('GeoJSON', outvector, 'streams', None, None, crs)
        node_feats, stream_feats = nodes_to_features(nodes, feats, tr)
        nodes_writer.write_geojson_features(node_feats)
        streams_writer.write_geojson_features(stream_feats)
        events_writer = io.VectorWriter('GeoJSON', outvector, 'events', None, None, crs)
        RainTool(io.VectorReader(outvector, 'nodes'), events_writer, rain).process()
    finally:
        comm.allreduce_max(0.0)
    return dict(outdir=outdir, vector=outvector, nlabels=nlabels, events=events_writer.filepath, nodes=nodes_writer.filepath,
                streams=streams_writer.filepath, pourpoints=pourpoint_writer.filepath)
