"""GPU: several row bands of one DEM (one band context each, driven by threads over the in-process ThreadComm on a
single MI355X) must reproduce the undivided raster bit for bit: fill, depths, no-flats fill, D8, accumulation,
bluespot labels (scipy numbering) and watersheds."""
import threading

import numpy as np
import pytest

import oracle
from _cases import fbm, fixtures

pytestmark = pytest.mark.gpu


def run_bands(dem, nbands):
    from malstroem_amd.distributed import BandPipeline, ThreadComm
    out = [None] * nbands
    err = []

    def work(comm):
        try:
            p = BandPipeline(comm, dem.shape, device=0)
            p.upload_dem(dem[p.row0:p.row0 + p.nrows])
            p.fill()
            p.noflat()
            p.flowdir()
            p.accum()
            n = p.label()
            p.watershed()
            out[comm.rank] = {k: p.download(k) for k in ("filled", "depths", "noflat", "flowdir", "accum", "labels", "watersheds")}
            out[comm.rank]["nlabels"] = n
            out[comm.rank]["short_diag"] = (p.short, p.diag)
            out[comm.rank]["exchanges"] = dict(p.exchanges)
            p.close()
        except Exception as e:  # pragma: no cover
            err.append(e)
            raise

    threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(nbands)]
    [t.start() for t in threads]
    [t.join(600) for t in threads]
    assert not err, err
    return out


@pytest.mark.parametrize("nbands", [2, 3])
def test_bands_match_reference_fixture(nbands):
    fx = fixtures()
    out = run_bands(fx["dtm"], nbands)
    assert np.array_equal(np.concatenate([o["filled"] for o in out]), fx["filled"])
    assert np.array_equal(np.concatenate([o["depths"] for o in out]), fx["depths"])
    assert np.array_equal(np.concatenate([o["noflat"] for o in out]), fx["filled_no_flats"])
    assert np.array_equal(np.concatenate([o["flowdir"] for o in out]), fx["flowdir_noflats"])
    assert out[0]["short_diag"] == (7.275957614183426e-12, 1.0289757937229989e-11)
    acc = np.concatenate([o["accum"] for o in out])
    assert acc.min() >= 1 and acc.max() == 11158 and acc.sum() == 3578615
    lab, n = oracle.connected_components(fx["depths"])
    assert out[0]["nlabels"] == n == 523 and np.array_equal(np.concatenate([o["labels"] for o in out]), lab)
    ws = lab.copy()
    oracle.watersheds_from_labels(fx["flowdir_noflats"], ws, 0)
    assert np.array_equal(np.concatenate([o["watersheds"] for o in out]), ws)


@pytest.mark.parametrize("nbands,h,w", [(2, 700, 450), (4, 1024, 1024), (4, 1000, 130)])
def test_bands_match_oracle(nbands, h, w):
    dem = fbm(h, w, beta=2.0, seed=21)
    out = run_bands(dem, nbands)
    filled = oracle.fill_terrain(dem)
    short, diag = oracle.minimum_safe_short_and_diag(dem)
    fnf = oracle.fill_terrain_no_flats(dem, short, diag)
    assert np.array_equal(np.concatenate([o["filled"] for o in out]), filled)
    assert np.array_equal(np.concatenate([o["noflat"] for o in out]), fnf)
    fd = oracle.terrain_flowdirection(fnf)
    assert np.array_equal(np.concatenate([o["flowdir"] for o in out]), fd)
    assert out[0]["short_diag"] == (short, diag)
    assert np.array_equal(np.concatenate([o["accum"] for o in out]), oracle.accumulated_flow(fd))
    lab, n = oracle.connected_components(oracle.depths(filled, dem))
    assert all(o["nlabels"] == n for o in out)
    assert np.array_equal(np.concatenate([o["labels"] for o in out]), lab)
    ws = lab.copy()
    oracle.watersheds_from_labels(fd, ws, 0)
    assert np.array_equal(np.concatenate([o["watersheds"] for o in out]), ws)
