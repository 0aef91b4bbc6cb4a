"""GPU: the no-flats fill as an integer geodesic distance transform (csrc/noflat_geo.hip) against the oracle
(reference fill.py:174-232, _fill.pyx:72-124), bit for bit, and the float64 relaxation it falls back to.

The integer path covers every level of the plain fill whose float64 binade gives integer weights for (short, diag); a raster
with a flat at elevation 0 (or NaN cells) runs the float64 relaxation instead.  Whatever path runs, its result has been checked
at every cell against the reference's equation on the device before the call returns -- these tests check it against the
oracle, and that the context reports the path that ran."""
import numpy as np
import pytest

import oracle
from _cases import fbm

pytestmark = pytest.mark.gpu


def run(dem, short=None, diag=None):
    if short is not None:      # user epsilons: the drop-in function (mhip_fill_noflat_f64); the context always uses the minimum safe ones
        from malstroem_amd.algorithms import fill
        return fill.fill_terrain_no_flats(dem, short, diag), None, None
    from malstroem_amd.pipeline import HydroPipeline
    with HydroPipeline(dem.shape) as pipe:
        pipe.upload("dem", dem)
        pipe.run("fill", "noflat")
        pipe.sync()
        why = {k: pipe.get_int("noflat_reject" + k) for k in ("", "_irregular", "_unreached", "_mismatch")}   # (diagnostics of a fall-back)
        return pipe.download("noflat"), pipe.get_int("noflat_algorithm"), (pipe.get_int("noflat_rounds"), why)


def check(dem, algorithm=None, short=None, diag=None):
    dem = np.ascontiguousarray(dem, dtype=np.float32)
    got, alg, rounds = run(dem, short, diag)
    s, d = (short, diag) if short is not None else oracle.minimum_safe_short_and_diag(dem)
    want = oracle.fill_terrain_no_flats(dem, s, d)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (len(bad), bad[:5].tolist(), got[tuple(bad[0])], want[tuple(bad[0])])
    if algorithm is not None:
        assert alg == algorithm, (alg, rounds)
    return alg


@pytest.mark.parametrize("h,w,beta,seed", [(3, 3, 2.0, 1), (5, 7, 2.0, 2), (63, 65, 2.0, 3), (64, 64, 2.0, 4), (188, 250, 2.0, 5),
                                           (700, 450, 3.0, 6), (1024, 1024, 2.0, 7), (125, 2000, 2.5, 8), (2000, 126, 2.5, 9)])
def test_positive_elevations_take_the_integer_path(h, w, beta, seed):
    dem = fbm(h, w, beta=beta, seed=seed) + np.float32(3.0)        # no level at 0
    check(dem, algorithm=2)


def test_levels_in_many_binades_negative_elevations_and_minus_powers_of_two():
    dem = fbm(900, 700, beta=2.0, seed=11)
    check(dem * 8 - 400, algorithm=2)                               # -400 .. 400: ten binades, both signs
    q = np.round(dem - 50.0)                                        # integer levels: natural flats at -32, -16, -8, -4, -2, -1, 1, 2, ...
    q[q == 0] = 0.5
    check(q, algorithm=2)
    check(dem * 1e-3 + 1e-3, algorithm=2)                           # small positive values: classes far below the DEM maximum
    check(dem * 1e4 + 7.0, algorithm=2)


@pytest.mark.parametrize("n,levels,smooth,seed", [(200, 4, 6, 21), (333, 9, 10, 22), (640, 3, 14, 23)])
def test_terraces_spill_cells_next_to_the_lower_lake(n, levels, smooth, seed):
    """Natural flats on a few levels that touch each other: the sources a flat cell is NOT adjacent to (walls of pass_wl) and the
    flat cells those sources feed on their own level (the seeds of ng_first) in every tile, across the tile seams (CPU model of the
    argument: test_noflat_walls_model.py)."""
    rng = np.random.default_rng(seed)
    z = rng.random((n, n))
    for _ in range(smooth):
        z = (z + np.roll(z, 1, 0) + np.roll(z, -1, 0) + np.roll(z, 1, 1) + np.roll(z, -1, 1)) / 5.0
    z = (z - z.min()) / (z.max() - z.min())
    check(np.floor(z * levels) + 3.0, algorithm=2)                  # one binade class (3 .. 3 + levels < 8 for the first case) ...
    check(np.floor(z * levels) * 3.0 + 1.0, algorithm=2)            # ... and several in a window


def test_quantised_terrain_large_natural_flats():
    dem = np.round(fbm(800, 800, beta=2.5, seed=12) / 2) * 2 + 10
    check(dem, algorithm=2)
    check(np.full((300, 200), 5.0), algorithm=2)                    # one flat, sources = the raster border
    stairs = np.repeat(np.arange(40, dtype=np.float32)[:, None], 500, 1) + 1
    check(np.repeat(stairs, 8, 0), algorithm=2)                     # terraces: every flat cell has a lower row within 8 cells


def test_a_flat_at_elevation_zero_is_left_to_the_float64_relaxation():
    """hybrid: the integer transform for every regular level, the relaxation only where the irregular flats lie; NaN cells send
    the whole raster to the relaxation"""
    dem = fbm(400, 300, beta=2.0, seed=13)
    dem[100:200, :250] = 0.0                                        # a sea at 0 that reaches the raster border: no constant ulp above 0
    check(dem, algorithm=3)
    sea = fbm(1500, 1200, beta=2.0, seed=17) - 30.0                 # a tenth of the raster below sea level ...
    sea[sea < 0] = 0.0                                              # ... clamped to 0: one huge irregular flat + regular lakes on land
    check(sea, algorithm=3)
    from _cases import fixtures
    fx = fixtures()                                                 # the reference's own DEM has cells at 0
    got, alg, _ = run(fx["dtm"])
    assert alg == 3 and np.array_equal(got, fx["filled_no_flats"])
    nan = fbm(300, 300, beta=2.0, seed=14) + 3
    nan[40:60, 70] = np.nan
    from malstroem_amd.algorithms import fill
    s, d = oracle.minimum_safe_short_and_diag(np.nan_to_num(nan))
    a, b = fill.fill_terrain_no_flats(nan, s, d), fill.fill_terrain_no_flats(nan, s, d)
    assert np.array_equal(a, b, equal_nan=True)                     # (NaN semantics are pinned in test_gpu_edgecases)


def test_a_flat_longer_than_the_uint32_headroom():
    """a 1-cell-wide channel at 0.3 m that snakes through walls of 100 m: 20 000 steps of 2**18 ulps each -- past 2**31 the
    distances are handed to the relaxation (every smaller one is exact), the result is the oracle's"""
    h, w = 203, 203
    dem = np.full((h, w), 100.0, np.float32)
    for k, r in enumerate(range(1, h - 1, 2)):
        dem[r, 1:w - 1] = 0.3
        if r + 2 < h - 1:
            dem[r + 1, (w - 2) if k % 2 == 0 else 1] = 0.3      # the turn at alternating ends
    dem[1, 0] = 0.2                                              # the outlet, on the raster border
    check(dem, algorithm=3)


@pytest.mark.parametrize("h,w,beta,seed", [(300, 260, 2.0, 5), (700, 513, 3.0, 6), (190, 1000, 2.5, 7)])
def test_self_listing_tail_rounds_on_small_rasters(monkeypatch, h, w, beta, seed):
    """At 16384^2 the rounds after the first batch hold few tiles and build the next round's list themselves (atomic appends
    instead of mark bytes + a compaction launch).  MHIP_NG_BATCH=1 makes every raster take that path from its second round on."""
    monkeypatch.setenv("MHIP_NG_BATCH", "1")
    dem = fbm(h, w, beta=beta, seed=seed) + np.float32(10.0)
    dem[h // 3:h // 3 + 40, w // 4:w // 4 + 150] = dem[h // 3, w // 4]            # a flat that spans several tiles
    assert check(dem) == 2


def test_the_device_check_is_live(monkeypatch):
    """MHIP_NG_CORRUPT makes the transform hand wrong distances (one raster row) to the final check: the check has to see them,
    the call has to fall back to the float64 relaxation (algorithm 0) -- and still return the reference's surface."""
    dem = fbm(300, 260, beta=2.0, seed=5) + np.float32(10.0)
    assert check(dem) == 2
    monkeypatch.setenv("MHIP_NG_CORRUPT", "1")
    assert check(dem) == 0


@pytest.mark.parametrize("short,diag", [(2.0 ** -30, 2.0 ** -30 * 2 ** 0.5), (1e-9, 1.5e-9), (1e-3, 1.4142e-3), (3e-7, 3e-7), (2.0 ** -20, 2.0 ** -19),
                                        (1e-13, 1.5e-13), (0.0, 0.0), (0.25, 0.5)])
def test_user_epsilons_whatever_path_they_take(short, diag):
    """fill_terrain_no_flats(dtm, short, diag) takes any epsilons (fill.py:174): weights that are no integers in a level's
    binade are rounded the way float64 rounds them -- or, on a tie / outside the uint32 headroom / epsilons so large that
    levels start to interact, the float64 relaxation runs.  The oracle is the judge in every case."""
    dem = fbm(500, 400, beta=2.0, seed=15) + 2
    check(dem, short=short, diag=diag)


def test_size_independent_properties_at_8192():
    """what no oracle run is needed for: the reference's equation at every interior cell (recomputed on the host in numpy),
    border = dem, and no flat left (every interior cell has a strictly lower neighbour)"""
    from _cases import fbm as f
    dem = f(8192, 8192, beta=2.0, seed=16) + 1
    got, alg, rounds = run(dem)
    assert alg == 2 and rounds[0] > 10, rounds
    s, d = oracle.minimum_safe_short_and_diag(dem)
    G = got
    assert np.array_equal(G[0], dem[0]) and np.array_equal(G[-1], dem[-1]) and np.array_equal(G[:, 0], dem[:, 0]) and np.array_equal(G[:, -1], dem[:, -1])
    c = G[1:-1, 1:-1]
    md = np.minimum(np.minimum(G[:-2, :-2], G[:-2, 2:]), np.minimum(G[2:, :-2], G[2:, 2:])) + d
    me = np.minimum(np.minimum(G[:-2, 1:-1], G[2:, 1:-1]), np.minimum(G[1:-1, :-2], G[1:-1, 2:])) + s
    assert np.array_equal(c, np.maximum(np.minimum(md, me), dem[1:-1, 1:-1].astype(np.float64)))
    lowest = np.minimum(np.minimum(np.minimum(G[:-2, :-2], G[:-2, 2:]), np.minimum(G[2:, :-2], G[2:, 2:])),
                        np.minimum(np.minimum(G[:-2, 1:-1], G[2:, 1:-1]), np.minimum(G[1:-1, :-2], G[1:-1, 2:])))
    assert np.all(lowest < c)


def _flowdir_of_request(dem):
    from malstroem_amd.pipeline import HydroPipeline
    with HydroPipeline(dem.shape) as pipe:
        pipe.upload("dem", dem)
        pipe.run("fill", "noflat", "flowdir")           # one request: the no-flats fill's finishing pass may write the directions
        pipe.sync()
        fused = pipe.download("flowdir")
        alg = pipe.get_int("noflat_algorithm")
        pipe.run("flowdir")                              # a request of its own: d8.hip on the resident surface
        pipe.sync()
        return fused, pipe.download("flowdir"), alg


def test_flow_directions_whichever_pass_writes_them(monkeypatch):
    """In one request with the no-flats fill the flow directions come out of the geodesic transform's finishing pass
    (ng_finish_kernel: d8_code on the 3 x 3 neighbourhood it holds anyway); a partial surface (a sea at 0), a surface the final
    check rejects, and a FLOWDIR request of its own go through d8.hip.  Same bits as the oracle every time."""
    regular = np.ascontiguousarray(fbm(700, 450, beta=2.5, seed=21) + np.float32(3.0))
    regular[300:340, 100:300] = regular[300, 100]                                   # a flat over several tiles
    sea = np.ascontiguousarray(fbm(600, 520, beta=2.0, seed=22) - np.float32(30.0))
    sea[sea < 0] = 0.0
    for dem, want_alg in ((regular, 2), (sea, 3)):
        s, d = oracle.minimum_safe_short_and_diag(dem)
        want = oracle.terrain_flowdirection(oracle.fill_terrain_no_flats(dem, s, d))
        fused, alone, alg = _flowdir_of_request(dem)
        assert alg == want_alg
        assert np.array_equal(fused, want) and np.array_equal(alone, want)
    monkeypatch.setenv("MHIP_NG_CORRUPT", "1")           # the finishing pass sees wrong distances: its directions must not be used
    s, d = oracle.minimum_safe_short_and_diag(regular)
    want = oracle.terrain_flowdirection(oracle.fill_terrain_no_flats(regular, s, d))
    fused, alone, alg = _flowdir_of_request(regular)
    assert alg == 0 and np.array_equal(fused, want) and np.array_equal(alone, want)
