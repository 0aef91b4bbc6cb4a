"""Raster I/O boundary -- the windowed counterpart of ``malstroem.io.RasterReader`` / ``RasterWriter`` (reference io.py:21-159).

The reference hands whole rasters to GDAL (``ReadAsArray`` / ``WriteArray``): at 65536 x 65536 that is 17 GB per float32
raster on the host before anything reaches a device.  Here a raster moves in row windows: ``RasterReader.read_window`` /
``iter_windows`` decode only the strips or tiles a window touches, ``HydroPipeline.upload_from`` / ``download_to`` stream them
to and from the resident rasters, ``RasterWriter.write_window`` appends tiles as rows arrive -- the host never holds more than
one window.  Same constructor arguments, attributes (``filepath``, ``transform``, ``crs``, ``nodata``, ``nodatasubst``) and
``read()`` / ``write(array)`` methods as the reference, so the tools take either kind of object.

Format: GeoTIFF, single band, classic or BigTIFF, little or big endian, strips or tiles, uncompressed or deflate, predictor
1 or 2, (u)int8/16/32 and float32/64 samples -- what GDAL writes with the reference's creation options
(``tiled=yes, compress=deflate, bigtiff=if_safer, predictor=2``, io.py:112,129-139), which is also what the writer here
produces.  GDAL itself is not needed (and not present in this image).  The CRS travels as the raw GeoKey tags of the file it
was read from; WKT strings cannot be translated without GDAL and are refused by the writer.

Vector side: ``VectorWriter`` / ``VectorReader`` keep the constructor and the ``write_geojson_features`` /
``read_geojson_features`` methods of the reference (io.py:160-330) but know ONE format, a GeoJSON FeatureCollection per layer
(``<datasource>/<layername>.geojson``) -- what the tools hand to each other between the stages of ``complete``.  OGR formats
stay with the reference's own classes, which the tools accept just as well.
"""
import json
import os
import struct
import zlib

import numpy as np

__all__ = ["RasterReader", "RasterWriter", "BandRasterWriter", "GeoKeys", "VectorReader", "VectorWriter"]

_TYPE_FMT = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d", 16: "Q", 17: "q", 18: "Q"}
_TAG_WIDTH, _TAG_HEIGHT, _TAG_BITS, _TAG_COMPRESSION, _TAG_PHOTOMETRIC = 256, 257, 258, 259, 262
_TAG_STRIP_OFFSETS, _TAG_SPP, _TAG_ROWS_PER_STRIP, _TAG_STRIP_COUNTS, _TAG_PLANAR, _TAG_PREDICTOR = 273, 277, 278, 279, 284, 317
_TAG_TILE_W, _TAG_TILE_H, _TAG_TILE_OFFSETS, _TAG_TILE_COUNTS, _TAG_SAMPLE_FORMAT = 322, 323, 324, 325, 339
_TAG_PIXEL_SCALE, _TAG_TIEPOINT, _TAG_TRANSFORMATION, _TAG_GEOKEYS, _TAG_GEODOUBLES, _TAG_GEOASCII, _TAG_NODATA = 33550, 33922, 34264, 34735, 34736, 34737, 42113


class GeoKeys(object):
    """The raw GeoTIFF key tags (34735 / 34736 / 34737) of a file: an opaque CRS that a writer copies verbatim."""

    def __init__(self, directory=None, doubles=None, ascii_=None):
        self.directory, self.doubles, self.ascii = directory, doubles, ascii_

    def __bool__(self):
        return self.directory is not None

    __nonzero__ = __bool__


def _sample_dtype(bits, fmt, order):
    kind = {1: "u", 2: "i", 3: "f"}.get(fmt)
    if kind is None or bits not in (8, 16, 32, 64) or (kind == "f" and bits < 32):
        raise NotImplementedError("TIFF sample format %r with %r bits" % (fmt, bits))
    return np.dtype("%s%s%d" % (order, kind, bits // 8))


class RasterReader(object):
    """Read a single-band GeoTIFF as a whole (``read``) or in row windows (``read_window``, ``iter_windows``).

    ``nodatasubst``: nodata cells are replaced by this value, with the reference's rule (io.py:69-71: only when the file's
    nodata value is truthy -- a nodata value of 0 is NOT substituted, exactly like there)."""

    def __init__(self, filepath, nodatasubst=None):
        self.filepath = filepath
        self.nodatasubst = nodatasubst
        self._fh = open(filepath, "rb")
        head = self._fh.read(16)
        self._order = {b"II": "<", b"MM": ">"}.get(head[:2])
        if self._order is None:
            raise ValueError("%s is not a TIFF file" % filepath)
        magic = struct.unpack(self._order + "H", head[2:4])[0]
        if magic == 42:
            self._big = False
            ifd = struct.unpack(self._order + "I", head[4:8])[0]
        elif magic == 43:
            self._big = True
            ifd = struct.unpack(self._order + "Q", head[8:16])[0]
        else:
            raise ValueError("%s is not a TIFF file" % filepath)
        t = self._tags = self._read_ifd(ifd)
        self.width, self.height = int(t[_TAG_WIDTH][0]), int(t[_TAG_HEIGHT][0])
        self.shape = (self.height, self.width)
        if int(t.get(_TAG_SPP, [1])[0]) != 1:
            raise NotImplementedError("only single band rasters")
        self.dtype = _sample_dtype(int(t[_TAG_BITS][0]), int(t.get(_TAG_SAMPLE_FORMAT, [1])[0]), self._order)
        self._compression = int(t.get(_TAG_COMPRESSION, [1])[0])
        if self._compression not in (1, 8, 32946):
            raise NotImplementedError("TIFF compression %d (only none / deflate)" % self._compression)
        self._predictor = int(t.get(_TAG_PREDICTOR, [1])[0])
        if self._predictor not in (1, 2):
            raise NotImplementedError("TIFF predictor %d" % self._predictor)
        if _TAG_TILE_W in t:
            self._bw, self._bh = int(t[_TAG_TILE_W][0]), int(t[_TAG_TILE_H][0])
            self._offsets, self._counts = t[_TAG_TILE_OFFSETS], t[_TAG_TILE_COUNTS]
        else:
            self._bw, self._bh = self.width, int(t.get(_TAG_ROWS_PER_STRIP, [self.height])[0])
            self._bh = min(self._bh, self.height)
            self._offsets, self._counts = t[_TAG_STRIP_OFFSETS], t[_TAG_STRIP_COUNTS]
        self._across = -(-self.width // self._bw)
        # georeferencing
        self.transform = None
        if _TAG_TRANSFORMATION in t:
            m = t[_TAG_TRANSFORMATION]
            self.transform = (m[3], m[0], m[1], m[7], m[4], m[5])
        elif _TAG_PIXEL_SCALE in t and _TAG_TIEPOINT in t:
            sx, sy = t[_TAG_PIXEL_SCALE][0], t[_TAG_PIXEL_SCALE][1]
            tp = t[_TAG_TIEPOINT]
            self.transform = (tp[3] - tp[0] * sx, sx, 0.0, tp[4] + tp[1] * sy, 0.0, -sy)
        self.crs = GeoKeys(t.get(_TAG_GEOKEYS), t.get(_TAG_GEODOUBLES), t.get(_TAG_GEOASCII))
        self.nodata = None
        if _TAG_NODATA in t:
            try:
                self.nodata = float(t[_TAG_NODATA].strip("\x00 "))
            except ValueError:
                self.nodata = None

    def close(self):
        self._fh.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _read_ifd(self, pos):
        o, fh = self._order, self._fh
        fh.seek(pos)
        if self._big:
            n = struct.unpack(o + "Q", fh.read(8))[0]
            raw = fh.read(20 * n)
            esize, cfmt, inline = 20, "HHQ", 8
        else:
            n = struct.unpack(o + "H", fh.read(2))[0]
            raw = fh.read(12 * n)
            esize, cfmt, inline = 12, "HHI", 4
        tags = {}
        for i in range(n):
            e = raw[i * esize:(i + 1) * esize]
            tag, typ, count = struct.unpack(o + cfmt, e[:esize - inline])
            fmt = _TYPE_FMT.get(typ)
            if fmt is None:
                continue
            size = struct.calcsize("=" + fmt) * count
            if size <= inline:
                data = e[esize - inline:esize - inline + size]
            else:
                off = struct.unpack(o + ("Q" if self._big else "I"), e[esize - inline:])[0]
                fh.seek(off)
                data = fh.read(size)
            if typ == 2:
                tags[tag] = data.decode("latin-1")
            else:
                vals = np.frombuffer(data, dtype=np.dtype(o + {"B": "u1", "H": "u2", "I": "u4", "II": "u4", "b": "i1", "h": "i2", "i": "i4", "ii": "i4",
                                                                 "f": "f4", "d": "f8", "Q": "u8", "q": "i8"}[fmt]))
                tags[tag] = vals.tolist()
        return tags

    def _block(self, index):
        """decoded block (strip or tile) ``index`` as a 2-D array of ``_bh`` x ``_bw`` (a last strip may be shorter)"""
        self._fh.seek(int(self._offsets[index]))
        raw = self._fh.read(int(self._counts[index]))
        if self._compression != 1:
            raw = zlib.decompress(raw)
        rows = len(raw) // (self._bw * self.dtype.itemsize)
        a = np.frombuffer(raw, dtype=self.dtype, count=rows * self._bw).reshape(rows, self._bw)
        if self._predictor == 2:
            u = a.view(np.dtype("%su%d" % (self._order, self.dtype.itemsize)))
            a = np.cumsum(u, axis=1, dtype=u.dtype).view(self.dtype)
        return a

    def read_window(self, row0, nrows):
        """Rows [row0, row0 + nrows) as a native-endian 2-D array; only the strips / tiles they touch are decoded."""
        row0, nrows = int(row0), int(nrows)
        if row0 < 0 or nrows < 0 or row0 + nrows > self.height:
            raise ValueError("window [%d, %d) outside the raster's %d rows" % (row0, row0 + nrows, self.height))
        out = np.empty((nrows, self.width), dtype=self.dtype.newbyteorder("="))
        for by in range(row0 // self._bh, -(-(row0 + nrows) // self._bh) if nrows else 0):
            top = by * self._bh
            a0, a1 = max(row0, top), min(row0 + nrows, top + self._bh, self.height)
            for bx in range(self._across):
                blk = self._block(by * self._across + bx)
                w = min(self._bw, self.width - bx * self._bw)
                out[a0 - row0:a1 - row0, bx * self._bw:bx * self._bw + w] = blk[a0 - top:a1 - top, :w]
        if self.nodata and self.nodatasubst is not None:           # io.py:69-71 (a nodata value of 0 is falsy there as well)
            mask = np.isnan(out) if np.isnan(self.nodata) else np.isclose(out, self.nodata)
            out[mask] = self.nodatasubst
        return out

    def read(self):
        """The whole raster as a 2-D array (reference io.py:60-73)."""
        return self.read_window(0, self.height)

    def iter_windows(self, max_rows=None):
        """(row0, array) windows covering the raster top to bottom; ``max_rows`` defaults to a multiple of the block height
        worth about 64 MiB."""
        if max_rows is None:
            per = max(1, (64 << 20) // max(1, self.width * self.dtype.itemsize))
            max_rows = max(self._bh, per // self._bh * self._bh)
        r = 0
        while r < self.height:
            n = min(int(max_rows), self.height - r)
            yield r, self.read_window(r, n)
            r += n


def _write_ifd(fh, big, shape, dtype, predictor, tile, offsets, counts, transform, crs, nodata):
    """The image file directory of a tiled, deflate-compressed single-band GeoTIFF at the current position of ``fh`` (out-of-line
    values first), and the pointer to it in the header."""
    H, W = shape
    kind = dtype.kind
    entries = []     # (tag, type, values)
    offtype = 16 if big else 4
    entries += [(_TAG_WIDTH, 4, [W]), (_TAG_HEIGHT, 4, [H]), (_TAG_BITS, 3, [dtype.itemsize * 8]), (_TAG_COMPRESSION, 3, [8]),
                (_TAG_PHOTOMETRIC, 3, [1]), (_TAG_SPP, 3, [1]), (_TAG_PLANAR, 3, [1]), (_TAG_PREDICTOR, 3, [predictor]),
                (_TAG_TILE_W, 3, [tile]), (_TAG_TILE_H, 3, [tile]), (_TAG_TILE_OFFSETS, offtype, offsets),
                (_TAG_TILE_COUNTS, offtype, counts), (_TAG_SAMPLE_FORMAT, 3, [{"u": 1, "i": 2, "f": 3}[kind]])]
    if transform:
        t = [float(v) for v in transform]
        if t[2] == 0.0 and t[4] == 0.0:
            entries += [(_TAG_PIXEL_SCALE, 12, [t[1], -t[5], 0.0]), (_TAG_TIEPOINT, 12, [0.0, 0.0, 0.0, t[0], t[3], 0.0])]
        else:
            entries.append((_TAG_TRANSFORMATION, 12, [t[1], t[2], 0.0, t[0], t[4], t[5], 0.0, t[3], 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]))
    if isinstance(crs, GeoKeys) and crs:
        entries.append((_TAG_GEOKEYS, 3, list(crs.directory)))
        if crs.doubles:
            entries.append((_TAG_GEODOUBLES, 12, list(crs.doubles)))
        if crs.ascii:
            entries.append((_TAG_GEOASCII, 2, crs.ascii))
    if nodata is not None:
        entries.append((_TAG_NODATA, 2, ("%.17g" % nodata) + "\x00"))
    entries.sort(key=lambda e: e[0])
    # out-of-line values first, then the IFD
    inline = 8 if big else 4
    packed = []
    for tag, typ, vals in entries:
        if typ == 2:
            data = vals.encode("latin-1")
            count = len(data)
        else:
            fmt = _TYPE_FMT[typ]
            data = struct.pack("<%d%s" % (len(vals), fmt), *vals)
            count = len(vals)
        if len(data) > inline:
            if fh.tell() & 1:
                fh.write(b"\x00")
            off = fh.tell()
            fh.write(data)
            data = struct.pack("<Q" if big else "<I", off)
        packed.append((tag, typ, count, data.ljust(inline, b"\x00")))
    if fh.tell() & 1:
        fh.write(b"\x00")
    ifd = fh.tell()
    if big:
        fh.write(struct.pack("<Q", len(packed)))
        for tag, typ, count, data in packed:
            fh.write(struct.pack("<HHQ", tag, typ, count) + data)
        fh.write(struct.pack("<Q", 0))
        fh.seek(8)
        fh.write(struct.pack("<Q", ifd))
    else:
        fh.write(struct.pack("<H", len(packed)))
        for tag, typ, count, data in packed:
            fh.write(struct.pack("<HHI", tag, typ, count) + data)
        fh.write(struct.pack("<I", 0))
        fh.seek(4)
        fh.write(struct.pack("<I", ifd))


_POOL = None


def _deflate_pool():
    """a few threads for the deflate of a tile row's tiles (one pool per process, made on first use)"""
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(2, min(8, (os.cpu_count() or 2) // 2)), thread_name_prefix="deflate")
    return _POOL


def _encode_tile(tile, predictor, itemsize, zlevel):
    if predictor == 2:
        u = tile.view(np.dtype("<u%d" % itemsize))
        d = u.copy()
        d[:, 1:] = u[:, 1:] - u[:, :-1]
        tile = d
    return zlib.compress(tile.tobytes(), zlevel)


class BandRasterWriter(object):
    """ONE GeoTIFF written by all ranks of a row-banded raster at the same time (reference: every tool writes the raster it holds,
    io.py:141-159, bluespots.py:175-214 -- here the raster is spread over the ranks, and none of them may become the funnel).

    Same format as ``RasterWriter`` (256 x 256 tiles, deflate, predictor 2 except for float64, always BigTIFF).  A rank writes
    the tile rows that START in its band; the rows of its last tile row that belong to the next band (< 256) come from that
    neighbour -- the only raster data that crosses ranks.  Compressed sizes are not known in advance, so every tile owns a SLOT
    of the worst-case deflate size at a position that follows from its index alone (``os.pwrite``: no rank waits for another); the
    unused tail of a slot is a hole of the (sparse) file, the tile-offset / byte-count tags say where the data is, like in any
    TIFF.  Rank 0 writes the header at the start and the directory behind the last slot once the byte counts are gathered (8
    bytes per tile).  Host memory per rank: one tile row of the raster + the neighbour's rows, whatever the raster's height.

    ``comm``: ``rank`` / ``size`` / ``exchange_rows`` / ``allgather`` / ``allreduce_max`` (a ``malstroem_amd.distributed.Comm``);
    ``rows(r0, n)``: rows [r0, r0 + n) of THIS rank's band (band-local) as an (n, W) array; ``extents``: (row0, nrows) of every
    rank, the same list on all of them.  Every rank calls ``write``; see ``supports`` for the band layouts it takes."""

    TILE = 256

    def __init__(self, filepath, transform, crs, nodata=None, zlevel=6):
        if isinstance(crs, str) and crs:
            raise NotImplementedError("a WKT coordinate system needs GDAL; pass the `crs` of a malstroem_amd.io.RasterReader (or None)")
        self.filepath, self.transform, self.crs, self.nodata, self.zlevel = filepath, transform, crs, nodata, int(zlevel)

    @classmethod
    def tile_rows(cls, extents, rank, H):
        """[k0, k1): the tile rows that start inside the band of ``rank``"""
        T = cls.TILE
        row0, nrows = extents[rank]
        k0 = -(-row0 // T)
        k1 = -(-(row0 + nrows) // T) if rank + 1 < len(extents) else -(-H // T)
        return k0, max(k0, k1)

    @classmethod
    def supports(cls, extents, H):
        """every tile row ends in the band it starts in or in the next one (true whenever the bands hold 255 rows or more)"""
        T = cls.TILE
        for r in range(len(extents) - 1):
            end = extents[r][0] + extents[r][1]
            if min(H, -(-end // T) * T) - end > extents[r + 1][1] or extents[r][1] < 1:
                return False
        return True

    def write(self, comm, shape, extents, rows, dtype):
        T = self.TILE
        H, W = int(shape[0]), int(shape[1])
        dtype = np.dtype(dtype).newbyteorder("<")
        me, size = comm.rank, comm.size
        row0, nrows = extents[me]
        if not self.supports(extents, H):
            raise ValueError("BandRasterWriter: a tile row (%d rows) must end in the band it starts in or in the next one" % T)
        predictor = 1 if dtype == np.dtype(np.float64) else 2
        across = -(-W // T)
        ntiles = across * -(-H // T)
        raw = T * T * dtype.itemsize
        slot = (raw + raw // 1000 + 64 + 15) & ~15                # (zlib's worst case is raw + 5 bytes per 16 KB block + 6)
        k0, k1 = self.tile_rows(extents, me, H)
        # rows of my band's head that complete the previous rank's last tile row / rows I need from the next rank
        give = min(nrows, k0 * T - row0) if me > 0 else 0
        need = max(0, min(H, k1 * T) - (row0 + nrows)) if me + 1 < size else 0
        to_up = np.ascontiguousarray(rows(0, give), dtype=dtype) if give else (np.zeros((1, W), dtype) if me > 0 else None)
        to_down = (np.zeros((max(need, 1), W), dtype) if me + 1 < size else None)   # (nothing travels down: the transports want equal shapes per seam)
        _, from_down = comm.exchange_rows(to_up, to_down) if size > 1 else (None, None)
        err = None
        try:
            if me == 0:
                with open(self.filepath, "wb") as fh:
                    fh.write(b"II" + struct.pack("<HHHQ", 43, 8, 0, 0))
        except OSError as e:
            err = e
        if comm.allreduce_max(1.0 if err else 0.0) > 0.0:          # (also: the file exists before anybody else opens it)
            raise err or IOError("rank 0 could not create %s" % self.filepath)
        counts = np.zeros((k1 - k0) * across, dtype=np.int64)
        try:
            fd = os.open(self.filepath, os.O_WRONLY)
            try:
                pending = np.zeros((T, across * T), dtype=dtype)
                for k in range(k0, k1):
                    g0, g1 = k * T, min(H, (k + 1) * T)               # global rows of this tile row
                    own1 = min(g1, row0 + nrows)
                    pending[:] = 0
                    if own1 > g0:
                        pending[:own1 - g0, :W] = rows(g0 - row0, own1 - g0)
                    if g1 > own1:                                     # the tail of my last tile row lives in the next band
                        pending[own1 - g0:g1 - g0, :W] = from_down[:g1 - own1]
                    tiles = [np.ascontiguousarray(pending[:, bx * T:(bx + 1) * T]) for bx in range(across)]
                    isz, zl = dtype.itemsize, self.zlevel
                    raws = list(_deflate_pool().map(lambda t: _encode_tile(t, predictor, isz, zl), tiles)) if across >= 4 else \
                        [_encode_tile(t, predictor, isz, zl) for t in tiles]
                    for bx, data in enumerate(raws):
                        assert len(data) <= slot
                        os.pwrite(fd, data, 16 + (k * across + bx) * slot)
                        counts[(k - k0) * across + bx] = len(data)
            finally:
                os.close(fd)
        except Exception as e:      # (voted on below: every rank raises or none does)
            err = e
        parts = comm.allgather((k0, counts))
        if comm.allreduce_max(1.0 if err else 0.0) > 0.0:
            raise err or IOError("another rank failed writing %s" % self.filepath)
        if me == 0:
            allc = np.zeros(ntiles, dtype=np.int64)
            for pk0, pc in parts:
                pc = np.asarray(pc, dtype=np.int64)
                allc[pk0 * across:pk0 * across + pc.size] = pc
            offsets = (16 + np.arange(ntiles, dtype=np.int64) * slot).tolist()
            with open(self.filepath, "r+b") as fh:
                fh.seek(16 + ntiles * slot)
                _write_ifd(fh, True, (H, W), dtype, predictor, T, offsets, allc.tolist(), self.transform, self.crs, self.nodata)
        comm.allreduce_max(0.0)                                      # (the file is complete when anybody returns)
        return self.filepath


class RasterWriter(object):
    """Write a 2-D array as a tiled, deflate-compressed GeoTIFF, at once (``write``) or in top-to-bottom row windows
    (``open`` / ``write_window`` / ``close``).  Creation options of the reference (io.py:112,129-139): 256 x 256 tiles,
    deflate, horizontal predictor for float32 / int32 / uint8 (not float64), BigTIFF when the raster may exceed 4 GB."""

    TILE = 256

    def __init__(self, filepath, transform, crs, nodata=None):
        if isinstance(crs, str) and crs:
            raise NotImplementedError("a WKT coordinate system needs GDAL; pass the `crs` of a malstroem_amd.io.RasterReader (or None)")
        self.filepath = filepath
        self.transform = transform
        self.crs = crs
        self.driver = 'gtiff'
        self.options = dict(tiled='yes', compress='deflate', bigtiff='if_safer', zlevel=6)
        self.nodata = nodata
        self._fh = None

    # ---- whole raster (reference API)
    def write(self, data):
        data = np.asarray(data)
        self.open(data.shape, data.dtype)
        self.write_window(0, data)
        self.close()

    # ---- windows
    def open(self, shape, dtype):
        dtype = np.dtype(dtype)
        if dtype not in (np.dtype(np.float64), np.dtype(np.float32), np.dtype(np.int32), np.dtype(np.uint8), np.dtype(np.uint16),
                         np.dtype(np.int16), np.dtype(np.uint32)):
            raise NotImplementedError("Cannot determine GeoTIFF datatype for numpy datatype {}".format(dtype))
        self._shape = (int(shape[0]), int(shape[1]))
        self._dtype = dtype.newbyteorder("<")
        self._predictor = 1 if dtype == np.float64 else 2
        self.options['predictor'] = self._predictor
        self._big = self.options.get('bigtiff') == 'yes' or (self.options.get('bigtiff') == 'if_safer' and
                                                              self._shape[0] * self._shape[1] * dtype.itemsize > 4000000000)
        self._fh = open(self.filepath, "wb")
        self._fh.write(b"II" + (struct.pack("<HHHQ", 43, 8, 0, 0) if self._big else struct.pack("<HI", 42, 0)))
        self._across = -(-self._shape[1] // self.TILE)
        self._offsets, self._counts = [], []
        self._pending = np.zeros((self.TILE, self._across * self.TILE), dtype=self._dtype)
        self._next_row = 0
        return self

    def _flush_tile_row(self, rows_valid):
        T = self.TILE
        if rows_valid < T:
            self._pending[rows_valid:] = 0
        # the tiles of a tile row are deflated side by side (zlib works without the GIL): the seven rasters of `complete` on a 4096^2
        # terrain spent 4 s in zlib one tile after the other; the file is the same, tile after tile in raster order
        zl, isz, pred = int(self.options.get('zlevel', 6)), self._dtype.itemsize, self._predictor
        tiles = [np.ascontiguousarray(self._pending[:, bx * T:(bx + 1) * T]) for bx in range(self._across)]
        if self._across >= 4:
            raws = list(_deflate_pool().map(lambda t: _encode_tile(t, pred, isz, zl), tiles))
        else:
            raws = [_encode_tile(t, pred, isz, zl) for t in tiles]
        for raw in raws:
            self._offsets.append(self._fh.tell())
            self._counts.append(len(raw))
            self._fh.write(raw)
            if len(raw) & 1:
                self._fh.write(b"\x00")

    def write_window(self, row0, data):
        """Rows [row0, row0 + len(data)): windows must arrive top to bottom without gaps."""
        data = np.asarray(data)
        if self._fh is None:
            raise ValueError("open(shape, dtype) first")
        if int(row0) != self._next_row or data.ndim != 2 or data.shape[1] != self._shape[1] or row0 + data.shape[0] > self._shape[0]:
            raise ValueError("windows must be full-width row blocks written in order (next row: %d)" % self._next_row)
        T, W = self.TILE, self._shape[1]
        done = 0
        while done < data.shape[0]:
            fill = self._next_row % T
            take = min(T - fill, data.shape[0] - done)
            self._pending[fill:fill + take, :W] = data[done:done + take]
            done += take
            self._next_row += take
            if self._next_row % T == 0 or self._next_row == self._shape[0]:
                self._flush_tile_row(((self._next_row - 1) % T) + 1)

    def close(self):
        if self._fh is None:
            return
        if self._next_row != self._shape[0]:
            self._fh.close()
            self._fh = None
            raise ValueError("raster incomplete: %d of %d rows written" % (self._next_row, self._shape[0]))
        _write_ifd(self._fh, self._big, self._shape, self._dtype, self._predictor, self.TILE, self._offsets, self._counts, self.transform,
                   self.crs, self.nodata)
        self._fh.close()
        self._fh = None


def _layer_path(datasource, layername):
    if datasource.lower().endswith((".geojson", ".json")):
        return datasource
    return os.path.join(datasource, "%s.geojson" % layername)


class VectorWriter(object):
    """Features of one layer as a GeoJSON FeatureCollection file.  Constructor and ``write_geojson_features`` as in the
    reference (io.py:160-296); ``driver`` must be 'GeoJSON' (or None), ``fields`` / ``geomtype`` / ``dsco`` / ``lco`` are
    kept as attributes and otherwise unused.  ``datasource`` is a directory (created on first write) or a ``.geojson`` path."""

    def __init__(self, driver, datasource, layername, fields, geomtype, crs, dsco=(), lco=()):
        if driver not in (None, "GeoJSON", "geojson"):
            raise NotImplementedError("malstroem_amd.io.VectorWriter writes GeoJSON only (OGR driver %r needs GDAL)" % (driver,))
        self.driver = "GeoJSON"
        self.datasource = datasource
        self.layername = layername
        self.fields = list(fields) if fields else []
        self.geomtype = geomtype
        self.crs = crs
        self.dsco = list(dsco)
        self.lco = list(lco)

    @property
    def filepath(self):
        return _layer_path(self.datasource, self.layername)

    def write_geojson_features(self, geojsonfeatures):
        """A sequence of GeoJSON features or a FeatureCollection dict (io.py:251-277)."""
        if not isinstance(geojsonfeatures, dict) or geojsonfeatures.get('type', None) == 'Feature':
            geojsonfeatures = dict(type="FeatureCollection", features=list(geojsonfeatures))
        feats = []
        for f in geojsonfeatures["features"]:
            g = dict(type="Feature", geometry=f.get("geometry"), properties=f.get("properties", {}))
            if "id" in f:
                g["id"] = f["id"]
            feats.append(g)
        path = self.filepath
        d = os.path.dirname(path)
        if d and not os.path.isdir(d):
            os.makedirs(d)
        # feature by feature through json.dumps -- the C encoder; json.dump(obj, fh) walks the whole collection in the pure-Python
        # chunk iterator: 29 of the 40 s of `complete` on a 4096^2 terrain with 238 000 bluespots.  The bytes are the ones json.dump wrote.
        with open(path, "w") as fh:
            fh.write('{"type": "FeatureCollection", "name": %s, "features": [' % json.dumps(self.layername))
            for k in range(0, len(feats), 4096):
                chunk = ", ".join(json.dumps(g, default=_json_scalar) for g in feats[k:k + 4096])
                fh.write((", " if k else "") + chunk)
            fh.write("]}")

    def close(self):
        pass


def _json_scalar(o):
    if isinstance(o, np.generic):
        return o.item()
    if isinstance(o, np.ndarray):
        return o.tolist()
    raise TypeError("%r is not JSON serializable" % type(o))


class VectorReader(object):
    """Reads the features a ``VectorWriter`` wrote (reference io.py:299-330: ``read_geojson_features``)."""

    def __init__(self, datasource, layername=None):
        self.datasource = datasource
        self.layername = layername

    def read_geojson_features(self):
        with open(_layer_path(self.datasource, self.layername)) as fh:
            return json.load(fh)["features"]
