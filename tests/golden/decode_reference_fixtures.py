#!/usr/bin/env python3
"""Decode the reference's golden rasters (``/root/reference/tests/data/*.tif``) into
``tests/golden/reference_fixtures.npz`` without GDAL.

The seven files are classic little-endian single-band TIFFs, deflate compressed
(compression 8), horizontal-differencing predictor (2), stored either as one 256x256
tile or as strips.  They are DATA of the reference's own test-suite
(``tests/data/fixtures.py:5-11``); only the decoded arrays are committed here.

Run (in the build container only; the reference never travels to the GPU box):

    python tests/golden/decode_reference_fixtures.py [/root/reference/tests/data]
"""
import struct
import sys
import zlib
from pathlib import Path

import numpy as np

TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 12: "d", 16: "Q"}
TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 16: 8}


def _read_ifd(buf):
    assert buf[:2] == b"II" and struct.unpack_from("<H", buf, 2)[0] == 42, "classic little-endian TIFF expected"
    (off,) = struct.unpack_from("<I", buf, 4)
    (n,) = struct.unpack_from("<H", buf, off)
    tags = {}
    for i in range(n):
        tag, typ, cnt, val = struct.unpack_from("<HHI4s", buf, off + 2 + 12 * i)
        size = TYPE_SIZE[typ] * cnt
        data = val[:size] if size <= 4 else buf[struct.unpack("<I", val)[0]:][:size]
        if typ == 3:
            v = list(struct.unpack("<%dH" % cnt, data))
        elif typ == 4:
            v = list(struct.unpack("<%dI" % cnt, data))
        elif typ == 12:
            v = list(struct.unpack("<%dd" % cnt, data))
        elif typ == 2:
            v = data.rstrip(b"\0").decode("ascii", "replace")
        else:
            v = data
        tags[tag] = v
    return tags


def decode_tiff(path):
    buf = Path(path).read_bytes()
    t = _read_ifd(buf)
    width, height = t[256][0], t[257][0]
    bits, fmt = t[258][0], t.get(339, [1])[0]
    assert t[259][0] == 8, "deflate expected"
    predictor = t.get(317, [1])[0]
    dtype = {(8, 1): np.uint8, (32, 2): np.int32, (32, 3): np.float32, (64, 3): np.float64}[(bits, fmt)]
    utype = {8: np.uint8, 32: np.uint32, 64: np.uint64}[bits]
    out = np.zeros((height, width), dtype=dtype)

    def unpredict(raw, rows, cols):
        a = np.frombuffer(raw, dtype=utype).reshape(rows, cols)
        if predictor == 2:
            a = np.cumsum(a, axis=1, dtype=utype)
        return a.view(dtype)

    if 322 in t:  # tiled
        tw, th = t[322][0], t[323][0]
        offs, cnts = t[324], t[325]
        k = 0
        for ty in range(0, height, th):
            for tx in range(0, width, tw):
                tile = unpredict(zlib.decompress(buf[offs[k]:offs[k] + cnts[k]]), th, tw)
                h, w = min(th, height - ty), min(tw, width - tx)
                out[ty:ty + h, tx:tx + w] = tile[:h, :w]
                k += 1
    else:  # strips
        rps = t.get(278, [height])[0]
        offs, cnts = t[273], t[279]
        for k, (o, c) in enumerate(zip(offs, cnts)):
            r0 = k * rps
            h = min(rps, height - r0)
            out[r0:r0 + h] = unpredict(zlib.decompress(buf[o:o + c]), h, width)
    meta = dict(pixel_scale=t.get(33550), tiepoint=t.get(33922), nodata=t.get(42113))
    return out, meta


def main():
    src = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/data")
    names = ["dtm", "filled", "depths", "filled_no_flats", "flowdir_noflats", "labelled", "wsheds"]
    arrays = {}
    for n in names:
        a, meta = decode_tiff(src / (n + ".tif"))
        arrays[n] = a
        print(n, a.dtype, a.shape, "min", a.min(), "max", a.max(), meta)
    # GDAL-style geotransform of the fixtures (same for all seven files)
    _, meta = decode_tiff(src / "dtm.tif")
    sx, sy = meta["pixel_scale"][0], meta["pixel_scale"][1]
    tp = meta["tiepoint"]
    arrays["geotransform"] = np.array([tp[3], sx, 0.0, tp[4], 0.0, -sy], dtype=np.float64)
    out = Path(__file__).resolve().parent / "reference_fixtures.npz"
    np.savez_compressed(out, **arrays)
    print("wrote", out, out.stat().st_size, "bytes")


if __name__ == "__main__":
    main()
