"""Decode the reference's own vector fixtures (tests/data/{pourpoints,nodes,streams}.json, used by its
tests/data/fixtures.py:13-15 and tests/test_raster_net.py) into one compact data file.

Data only: feature properties and coordinates, no code.  Run once in the build container:
    python tests/golden/extract_reference_vectors.py [/root/reference/tests/data]
"""
import gzip
import json
import sys
from pathlib import Path


def main():
    src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/data")
    out = {}
    for name in ("pourpoints", "nodes", "streams"):
        d = json.loads((src / (name + ".json")).read_text())
        out[name] = [dict(properties={k: v for k, v in f["properties"].items() if k != "type"},
                          coordinates=f["geometry"]["coordinates"]) for f in d["features"]]
        print(name, len(out[name]), "features")
    path = Path(__file__).resolve().parent / "reference_vectors.json.gz"
    with gzip.open(path, "wt") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", path, path.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
