"""RainTool -- mirror of ``malstroem.rain.RainTool`` (reference rain.py:20-87): rain events on the node network.

Same constructor and ``process()`` contract: reads nodes from a vector reader (``read_geojson_features``), adds the columns
``rainv_{mm:g}``, ``spillv_{mm:g}``, ``v_{mm:g}``, ``pctv_{mm:g}`` for every event and writes the features back.  All events
are evaluated in one call of the library.
"""
import logging

from .network import Network


class RainTool(object):
    def __init__(self, input_nodes, output_eventdata, events_rainmm):
        self.input_nodes = input_nodes
        self.output_eventdata = output_eventdata
        self.events_rainmm = list(events_rainmm)
        self.logger = logging.getLogger(__name__)

    def process(self):
        self.logger.info("Reading input nodes")
        features = {f['properties']['nodeid']: f for f in self.input_nodes.read_geojson_features()}
        self.logger.info("Creating stream network")
        network = Network()
        network.add_nodes([f['properties'] for f in features.values()])
        self.logger.info("Calculating rain events")
        for mmrain, eventvalues in zip(self.events_rainmm, network.rain_events(self.events_rainmm)):
            self.logger.info("  {}mm".format(mmrain))
            for e in eventvalues:
                props = features[e['nodeid']]['properties']
                for key in ('rainv', 'spillv', 'v', 'pctv'):
                    props[self._output_property(key, mmrain)] = e[key]
        self.logger.info("Writing output")
        self.output_eventdata.write_geojson_features(features.values())
        self.logger.info("Done")

    def _output_property(self, string, mmrain):
        return "{}_{:g}".format(string, mmrain)
