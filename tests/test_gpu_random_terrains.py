"""GPU: randomised end-to-end parity -- the whole chain (one context, and 2-5 row bands) against the oracle on seeded random
terrains: shapes from 3 x 3 to 700 x 700, roughness, quantised elevations (natural flats), offsets into negative elevations, seas
at elevation 0, tiny and huge value ranges, plateaus.  Whatever engines run (tiled priority-flood / iterative schedule; integer
geodesic transform / hybrid / float64 relaxation), every raster equals the oracle's bit for bit."""
import threading

import numpy as np
import pytest

import oracle
from _cases import fbm

pytestmark = pytest.mark.gpu


def terrain(rng, h, w):
    dem = fbm(h, w, beta=float(rng.choice([1.5, 2.0, 2.5, 3.0])), seed=int(rng.integers(1 << 30)))
    mode = int(rng.integers(0, 7))
    if mode == 1:
        dem = np.round(dem / rng.choice([0.5, 1, 2, 5]))
    elif mode == 2:
        dem = dem - np.float32(rng.uniform(10, 90))
    elif mode == 3:
        dem = dem - np.float32(rng.uniform(10, 60))
        dem[dem < 0] = 0
    elif mode == 4:
        dem = dem * np.float32(rng.choice([1e-3, 1e3, 1e5]))
    elif mode == 5:
        dem = np.maximum(dem, np.float32(rng.uniform(20, 60)))
    elif mode == 6:
        dem = dem + rng.normal(0, 1e-3, dem.shape)
    return np.ascontiguousarray(dem, np.float32), mode


def expected(dem):
    filled = oracle.fill_terrain(dem)
    s, d = oracle.minimum_safe_short_and_diag(dem)
    fnf = oracle.fill_terrain_no_flats(dem, s, d)
    fd = oracle.terrain_flowdirection(fnf)
    lab, n = oracle.connected_components(oracle.depths(filled, dem))
    ws = lab.copy()
    oracle.watersheds_from_labels(fd, ws, 0)
    return dict(filled=filled, noflat=fnf, flowdir=fd, accum=oracle.accumulated_flow(fd), labels=lab, watersheds=ws), n


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_one_context(seed):
    from malstroem_amd.pipeline import HydroPipeline
    rng = np.random.default_rng(seed)
    engines = set()
    for it in range(14):
        dem, mode = terrain(rng, int(rng.integers(3, 700)), int(rng.integers(3, 700)))
        with HydroPipeline(dem.shape) as p:
            p.upload("dem", dem)
            p.run("fill", "noflat", "flowdir", "accum", "label")
            n = p.apply_keep(None)
            p.run("watershed", "pourpoints")
            p.sync()
            got = {k: p.download(k) for k in ("filled", "noflat", "flowdir", "accum", "labels", "watersheds")}
            engines.add((p.get_int("fill_algorithm"), p.get_int("noflat_algorithm")))
        want, on = expected(dem)
        assert n == on, (seed, it, dem.shape, mode)
        for k, v in want.items():
            assert np.array_equal(got[k], v), (seed, it, dem.shape, mode, k)
    assert (1, 2) in engines


@pytest.mark.parametrize("seed", [4, 5])
def test_row_bands(seed):
    from malstroem_amd.distributed import BandPipeline, ThreadComm
    rng = np.random.default_rng(seed)
    for it in range(6):
        nb = int(rng.integers(2, 6))
        dem, mode = terrain(rng, int(rng.integers(max(8, nb), 900)), int(rng.integers(3, 500)))
        out, err = [None] * nb, []

        def work(comm):
            try:
                p = BandPipeline(comm, dem.shape, device=0, align=bool(it % 3))
                p.upload_dem(dem[p.row0:p.row0 + p.nrows])
                p.run_chain(overlap=bool(it % 2))
                out[comm.rank] = {k: p.download(k) for k in ("filled", "noflat", "flowdir", "accum", "labels", "watersheds")}
                p.close()
            except Exception as e:  # pragma: no cover
                err.append(e)
                raise

        threads = [threading.Thread(target=work, args=(c,)) for c in ThreadComm.world(nb)]
        [t.start() for t in threads]
        [t.join(300) for t in threads]
        assert not err and all(o is not None for o in out), (seed, it, err)
        want, _ = expected(dem)
        for k, v in want.items():
            assert np.array_equal(np.concatenate([o[k] for o in out]), v), (seed, it, dem.shape, nb, mode, k)
