"""Raster I/O boundary: malstroem_amd.io's windowed GeoTIFF reader / writer.

CPU: round trips over every dtype / layout / size class, window reads against whole reads, nodata substitution with the
reference's rule (io.py:69-71), and -- where the reference checkout is present -- its seven tests/data/*.tif files decoded by
the product reader against the committed fixtures (decoded independently by tests/golden/decode_reference_fixtures.py).
GPU: DemTool driven by a windowed reader and windowed writers (one window on the host at a time) against the fixtures.
"""
import os

import numpy as np
import pytest

from _cases import fixtures
from malstroem_amd.io import RasterReader, RasterWriter

REF_DATA = "/root/reference/tests/data"


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.uint8, np.uint16])
@pytest.mark.parametrize("shape", [(1, 1), (37, 300), (256, 256), (257, 513), (700, 90)])
def test_writer_reader_round_trip(tmp_path, dtype, shape):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    a = (rng.standard_normal(shape) * 100).astype(dtype) if np.dtype(dtype).kind == "f" else rng.integers(0, 200, size=shape).astype(dtype)
    transform = (720000.0, 16.0, 0.0, 6193000.0, 0.0, -15.957446808510639)
    path = str(tmp_path / "a.tif")
    w = RasterWriter(path, transform, None, nodata=-999.0)
    w.write(a)
    assert w.options["predictor"] == (1 if dtype == np.float64 else 2) and w.options["compress"] == "deflate"   # io.py:112,129-139
    with RasterReader(path) as r:
        assert r.shape == shape and r.dtype == np.dtype(dtype) and r.nodata == -999.0
        assert np.allclose(r.transform, transform, rtol=0, atol=1e-9)
        assert np.array_equal(r.read(), a)
        for row0, n in ((0, 1), (shape[0] // 2, shape[0] - shape[0] // 2), (max(0, shape[0] - 3), min(3, shape[0]))):
            assert np.array_equal(r.read_window(row0, n), a[row0:row0 + n])
        got = np.concatenate([w_ for _, w_ in r.iter_windows(100)])
        assert np.array_equal(got, a)
        with pytest.raises(ValueError):
            r.read_window(shape[0], 1)
    # windowed writing in odd chunk sizes gives the same file contents
    path2 = str(tmp_path / "b.tif")
    w2 = RasterWriter(path2, transform, None).open(shape, dtype)
    row = 0
    for n in (1, 255, 2, 300, 10 ** 6):
        n = min(n, shape[0] - row)
        if n <= 0:
            break
        w2.write_window(row, a[row:row + n])
        row += n
    w2.close()
    with RasterReader(path2) as r:
        assert np.array_equal(r.read(), a)


def test_bigtiff_and_crs_pass_through(tmp_path):
    a = np.arange(300 * 270, dtype=np.float32).reshape(300, 270)
    path = str(tmp_path / "big.tif")
    w = RasterWriter(path, (0.0, 1.0, 0.0, 300.0, 0.0, -1.0), None)
    w.options["bigtiff"] = "yes"
    w.write(a)
    with open(path, "rb") as fh:
        assert fh.read(4) == b"II+\x00"        # BigTIFF magic 43
    with RasterReader(path) as r:
        assert np.array_equal(r.read(), a)
    with pytest.raises(NotImplementedError):
        RasterWriter(path, None, 'PROJCS["ETRS89 / UTM zone 32N"]')
    w = RasterWriter(str(tmp_path / "c.tif"), None, None).open((4, 4), np.uint8)
    w.write_window(0, np.zeros((2, 4), np.uint8))
    with pytest.raises(ValueError):
        w.write_window(3, np.zeros((1, 4), np.uint8))      # gap
    with pytest.raises(ValueError):
        w.close()                                           # incomplete


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int32, np.uint8])
@pytest.mark.parametrize("nbands,shape", [(3, (1300, 700)), (4, (1100, 256)), (2, (517, 90))])
def test_band_writer_all_ranks_write_one_file(tmp_path, dtype, nbands, shape):
    """io.BandRasterWriter: every rank of a row-banded raster writes the tile rows that start in its band into ONE GeoTIFF (slots of
    worst-case size, `os.pwrite`, the directory by rank 0); only the < 256 rows that complete a rank's last tile row come from its
    neighbour.  The file decodes to the raster, with the same tags as the one RasterWriter writes -- seams on and off the tile grid."""
    import threading
    from malstroem_amd.distributed import ThreadComm, band_rows
    from malstroem_amd.io import BandRasterWriter
    H, W = shape
    rng = np.random.default_rng(H + W)
    a = (np.cumsum(rng.random((H, W)), axis=1) * 3).astype(dtype)
    gt = (10.0, 2.0, 0.0, 50.0, 0.0, -2.0)
    RasterWriter(str(tmp_path / "one.tif"), gt, None, 0).write(a)
    ext = [band_rows(H, nbands, r, True) for r in range(nbands)]
    seen, errs = [], []

    def run(comm):
        try:
            r0, nr = ext[comm.rank]

            def rows(q0, q):
                assert 0 <= q0 and q0 + q <= nr
                seen.append((comm.rank, q))
                return a[r0 + q0:r0 + q0 + q]
            BandRasterWriter(str(tmp_path / "bands.tif"), gt, None, 0).write(comm, (H, W), ext, rows, dtype)
        except Exception as e:      # pragma: no cover
            import traceback
            traceback.print_exc()
            errs.append(e)
    threads = [threading.Thread(target=run, args=(c,), daemon=True) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(120) for t in threads]
    assert not errs and not any(t.is_alive() for t in threads)
    assert max(q for _, q in seen) <= 256                      # a rank never asks its band for more than one tile row
    with RasterReader(str(tmp_path / "one.tif")) as r1, RasterReader(str(tmp_path / "bands.tif")) as r2:
        got = r2.read()
        assert got.dtype == a.dtype and np.array_equal(got, a) and np.array_equal(r1.read(), a)
        assert r1.transform == r2.transform and r1.nodata == r2.nodata
        assert np.array_equal(r2.read_window(H // 2, 40), a[H // 2:H // 2 + 40])


def test_band_writer_refuses_a_tile_row_over_three_bands(tmp_path):
    from malstroem_amd.distributed import ThreadComm
    from malstroem_amd.io import BandRasterWriter
    c = ThreadComm.world(3)[0]
    with pytest.raises(ValueError):
        BandRasterWriter(str(tmp_path / "x.tif"), None, None).write(c, (300, 64), [(0, 100), (100, 100), (200, 100)], lambda a, b: None, np.float32)


def test_nodata_substitution_follows_the_reference_rule(tmp_path):
    a = np.array([[1.0, -9999.0, 3.0], [0.0, 5.0, -9999.0]], dtype=np.float32)
    p1, p2 = str(tmp_path / "n1.tif"), str(tmp_path / "n0.tif")
    RasterWriter(p1, None, None, nodata=-9999.0).write(a)
    RasterWriter(p2, None, None, nodata=0.0).write(a)
    assert np.array_equal(RasterReader(p1, nodatasubst=-999).read(), np.array([[1, -999, 3], [0, 5, -999]], np.float32))
    assert np.array_equal(RasterReader(p1).read(), a)
    assert np.array_equal(RasterReader(p2, nodatasubst=-999).read(), a)      # nodata == 0 is falsy in io.py:69: not substituted


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference checkout not on this box")
def test_reader_decodes_the_reference_fixtures():
    fx = fixtures()
    for name in ("dtm", "filled", "depths", "filled_no_flats", "flowdir_noflats", "labelled", "wsheds"):
        with RasterReader(os.path.join(REF_DATA, name + ".tif")) as r:
            a = r.read()
            assert a.dtype == fx[name].dtype and np.array_equal(a, fx[name]), name
            assert np.allclose(r.transform, fx["geotransform"], rtol=0, atol=1e-9)
            assert np.array_equal(r.read_window(100, 50), fx[name][100:150])
            assert r.crs        # the GeoKey tags travel as an opaque CRS


@pytest.mark.gpu
def test_demtool_streams_through_windowed_io(tmp_path):
    """DemTool with a windowed reader and windowed writers: rasters move device <-> file window by window; the files hold
    the reference's golden rasters."""
    from malstroem_amd.dem import DemTool
    fx = fixtures()
    gt = tuple(float(v) for v in fx["geotransform"])
    src = str(tmp_path / "dtm.tif")
    RasterWriter(src, gt, None).write(fx["dtm"])
    reader = RasterReader(src)
    outs = {k: RasterWriter(str(tmp_path / (k + ".tif")), gt, reader.crs) for k in ("filled", "flowdir", "depths", "accum")}
    DemTool(reader, outs["filled"], outs["flowdir"], outs["depths"], outs["accum"]).process()
    for k, want in (("filled", fx["filled"]), ("flowdir", fx["flowdir_noflats"]), ("depths", fx["depths"])):
        with RasterReader(str(tmp_path / (k + ".tif"))) as r:
            assert np.array_equal(r.read(), want), k
            assert np.allclose(r.transform, gt, rtol=0, atol=1e-9)
    with RasterReader(str(tmp_path / "accum.tif")) as r:
        acc = r.read()
        assert acc.dtype == np.float64 and acc.max() == 11158 and acc.sum() == 3578615
