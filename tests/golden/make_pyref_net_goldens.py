"""Golden vectors for the stream network, generated in the build container by IMPORTING the reference's
unmodified pure-Python modules from /root/reference (malstroem.algorithms.net, malstroem.network) and running them on the
reference's own fixtures.  Only inputs and outputs are stored (tests/golden/pyref_net.json.gz); no reference code travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_pyref_net_goldens.py
"""
import gzip
import json
import sys
from pathlib import Path

import numpy as np

sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
sys.path.insert(0, "/root/reference")
sys.path.insert(0, str(HERE.parent))

from malstroem.algorithms import net          # noqa: E402  (the reference)
from malstroem.network import Network         # noqa: E402  (the reference)
from _cases import fixtures, reference_vectors  # noqa: E402


def clean(nodes):
    out = []
    for n in nodes:
        d = dict(id=int(n['id']), downstream_id=None if n['downstream_id'] is None else int(n['downstream_id']), nodetype=n['nodetype'],
                 pix=[int(n['pix'][0]), int(n['pix'][1])])
        if 'geometry' in n:
            d['geometry'] = [[int(c[0]), int(c[1])] for c in n['geometry']]
        out.append(d)
    return out


def main():
    fx, vec = fixtures(), reference_vectors()
    fd, lab = fx["flowdir_noflats"], fx["labelled"]
    pix = [(p["properties"]["cell_row"], p["properties"]["cell_col"]) for p in vec["pourpoints"]]
    out = {}
    singles = []
    for cell in pix[1:21]:     # the 20 pour points of reference tests/test_raster_net.py:8-25
        for bg in (0, None):
            lbl, geom = net.next_downstream_label(fd, lab, cell, background_label=bg, geometry=True)
            singles.append(dict(cell=list(cell), background=bg, label=lbl, geometry=[[int(c[0]), int(c[1])] for c in geom]))
    out["next_downstream_label"] = singles
    out["pourpoint_network"] = clean(net.pourpoint_network(fd, lab, pix, 0))
    out["geometric_pourpoint_network"] = clean(net.geometric_pourpoint_network(fd, lab, pix, 0))
    # a second labelled raster (the watersheds) exercises labels everywhere and no background
    out["pourpoint_network_on_watersheds"] = clean(net.pourpoint_network(fd, fx["wsheds"], pix, None))
    nodes = [dict(n["properties"]) for n in vec["nodes"]]
    events = [10, 30, 100.5, 0]
    rain = {}
    for mm in events:
        nw = Network()
        nw.add_nodes([dict(n) for n in nodes])
        rain["%g" % mm] = nw.rain_event(mm)
    out["rain_nodes"] = nodes
    out["rain_events"] = rain
    path = HERE / "pyref_net.json.gz"
    with gzip.open(path, "wt") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("nodes:", len(out["geometric_pourpoint_network"]), "junctions:",
          sum(n["nodetype"] == "junction" for n in out["geometric_pourpoint_network"]), "->", path, path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
