"""StreamTool -- mirror of ``malstroem.streams.StreamTool`` (reference streams.py:20-124): upstream / downstream relations
between pour points, with junction nodes and stream geometries when ``output_streams`` is given.

``pipeline``: an optional ``HydroPipeline`` that still holds the flow directions and the filtered bluespot labels
(``DemTool`` -> ``BluespotTool`` on one pipeline): the walk then runs on the resident rasters and nothing is read again.
"""
import logging

from .algorithms import net
from .bluespots import transform_cell_to_world


class StreamTool(object):
    def __init__(self, input_pourpoints, input_bluespots, input_flowdir, output_nodes, output_streams=None, pipeline=None):
        self.input_pourpoints = input_pourpoints
        self.input_bluespots = input_bluespots
        self.input_flowdir = input_flowdir
        self.output_nodes = output_nodes
        self.output_streams = output_streams
        self.pipeline = pipeline
        self.logger = logging.getLogger(__name__)

    def process(self):
        self.logger.info("Read input data")
        transform = self.input_flowdir.transform
        pourpoints = self.input_pourpoints.read_geojson_features()
        pourpoints_pix = [(pp['properties']['cell_row'], pp['properties']['cell_col']) for pp in pourpoints]
        flowdir = labeled = None
        if self.pipeline is None:
            flowdir = self.input_flowdir.read()
            labeled = self.input_bluespots.read()
        self.logger.info("Processing stream network")
        if self.output_streams is not None:
            nodes = net.geometric_pourpoint_network(flowdir, labeled, pourpoints_pix, 0, pipeline=self.pipeline)
        else:
            nodes = net.pourpoint_network(flowdir, labeled, pourpoints_pix, 0, pipeline=self.pipeline)
        self.logger.info("Writing {} nodes".format(len(nodes)))
        geojson_nodes, streams = nodes_to_features(nodes, pourpoints, transform)
        self.output_nodes.write_geojson_features(geojson_nodes)
        if self.output_streams:
            self.output_streams.write_geojson_features(streams)


def nodes_to_features(nodes, pourpoints, transform):
    """(node features, stream features) of the reference's StreamTool output (streams.py:77-124): a node carries the bluespot
    attributes of its pour point (defaults for a junction), a stream the path from the node to the next one downstream."""
    pp_index = {pp['properties']['bspot_id']: pp for pp in pourpoints}
    geojson_nodes, streams = [], []
    for n in nodes:
        props = dict(nodeid=n['id'], dstrnodeid=n['downstream_id'], nodetype=n['nodetype'], cell_row=n['pix'][0], cell_col=n['pix'][1],
                     bspot_id=None, bspot_area=0.0, bspot_vol=0.0, wshed_area=0.0)      # defaults of a junction node
        ppoint = pp_index.get(n['id'], None)
        if ppoint:
            for key in ('bspot_id', 'bspot_area', 'bspot_vol', 'wshed_area'):
                props[key] = ppoint['properties'][key]
        coord = transform_cell_to_world(n['pix'], transform)
        geojson_nodes.append(dict(id=n['id'], geometry=dict(type='Point', coordinates=list(coord)), properties=props))
        if n.get('geometry'):
            coords = [transform_cell_to_world(c, transform) for c in n['geometry']]
            streams.append(dict(id=n['id'], geometry=dict(type='LineString', coordinates=list(coords)),
                                properties=dict(nodeid=n['id'], dstrnodeid=n['downstream_id'])))
    return geojson_nodes, streams
