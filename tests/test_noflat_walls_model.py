"""CPU: the argument behind the no-flats rounds' bit-mask passes (noflat_geo.hip, pass_wl), as a numpy model.

The geodesic distance of a flat cell is relaxed from the neighbours it is ADJACENT to (same level).  The kernel keeps no adjacency bits
for a window of one class: a flat cell has no lower neighbour, so a neighbour on another level is higher, has the flat cell as a lower
neighbour and is a source (distance 0, fixed).  Such a source becomes a WALL ("not reached") in the window, the flat cells of its own
level that it feeds start from their step weight (the seeds), and every flat cell then takes the minimum over all 8 neighbours.  The model
runs both formulations to their fixed points on random terraced surfaces and compares them cell by cell."""
import numpy as np
import pytest

INF = np.int64(1) << 40
S, G = 5, 7                                                   # straight / diagonal step (any positive integers)
NB = [(-1, 0, S), (-1, 1, G), (0, 1, S), (1, 1, G), (1, 0, S), (1, -1, G), (0, -1, S), (-1, -1, G)]


def shifted(a, dr, dc, fill):
    out = np.full_like(a, fill)
    H, W = a.shape
    out[max(0, -dr):H - max(0, dr), max(0, -dc):W - max(0, dc)] = a[max(0, dr):H - max(0, -dr), max(0, dc):W - max(0, -dc)]
    return out


def classify(F):
    H, W = F.shape
    interior = np.zeros((H, W), bool)
    interior[1:-1, 1:-1] = True
    lower = np.zeros((H, W), bool)
    equal = np.zeros((H, W), bool)
    for dr, dc, _ in NB:
        n = shifted(F, dr, dc, np.inf)
        lower |= n < F
        equal |= n == F
    src = lower | ~interior                                    # a lower neighbour, or the raster border
    flat = interior & ~lower & equal
    return src, flat


def fixed_point(D, movable, step):
    while True:
        new = step(D)
        new = np.where(movable, np.minimum(D, new), D)
        if (new == D).all():
            return D
        D = new


def masked_distances(F):
    """the definition: candidates only from neighbours of the same level"""
    src, flat = classify(F)
    D = np.where(flat, INF, 0).astype(np.int64)

    def step(D):
        best = np.full_like(D, INF)
        for dr, dc, w in NB:
            adj = shifted(F, dr, dc, np.inf) == F
            best = np.minimum(best, np.where(adj, shifted(D, dr, dc, INF) + w, INF))
        return best
    return fixed_point(D, flat, step), flat


def wall_distances(F):
    """the kernel's formulation: walls + seeds, then candidates from all 8 neighbours"""
    src, flat = classify(F)
    wall = np.zeros_like(flat)
    for dr, dc, _ in NB:                                        # a flat neighbour on another level
        wall |= shifted(flat, dr, dc, False) & (shifted(F, dr, dc, np.inf) != F)
    assert not (wall & flat).any()                              # (two flat cells next to each other are on one level)
    assert (wall <= src).all()                                  # a wall is a source
    D = np.where(flat, INF, 0).astype(np.int64)
    for dr, dc, w in NB:                                        # the seeds
        fed = flat & shifted(wall, dr, dc, False) & (shifted(F, dr, dc, np.inf) == F)
        D = np.where(fed, np.minimum(D, w), D)
    D = np.where(wall, INF, D)

    def step(D):
        best = np.full_like(D, INF)
        for dr, dc, w in NB:
            best = np.minimum(best, shifted(D, dr, dc, INF) + w)
        return best
    D = fixed_point(D, flat, step)
    return np.where(wall, 0, D), flat, int(wall.sum())


def terraces(rng, n, levels, smooth):
    z = rng.random((n, n))
    for _ in range(smooth):
        z = (z + np.roll(z, 1, 0) + np.roll(z, -1, 0) + np.roll(z, 1, 1) + np.roll(z, -1, 1)) / 5.0
    z = (z - z.min()) / (z.max() - z.min() + 1e-12)
    return np.floor(z * levels).astype(np.float32)


@pytest.mark.parametrize("seed,n,levels,smooth", [(1, 24, 3, 2), (2, 40, 5, 4), (3, 40, 12, 6), (4, 64, 4, 8), (5, 64, 30, 3), (6, 33, 2, 1)])
def test_walls_and_seeds_give_the_adjacency_masked_distances(seed, n, levels, smooth):
    F = terraces(np.random.default_rng(seed), n, levels, smooth)
    ref, flat = masked_distances(F)
    got, flat2, nwalls = wall_distances(F)
    assert (flat == flat2).all() and flat.sum() > n
    assert nwalls > 0                                           # the case the kernel's seeds exist for occurs
    np.testing.assert_array_equal(got[flat], ref[flat])
    assert (got[~flat] == 0).all()


def test_a_spill_cell_next_to_the_lower_lake():
    """the cascade: lake at 5 whose spill cell (a source of level 5) touches a flat cell of the lake at 3"""
    F = np.full((9, 12), 9.0, np.float32)
    F[2:7, 1:5] = 5.0                                           # upper lake ...
    F[4, 5] = 5.0                                               # ... its spill cell: lower neighbour (4, 6) -- a source
    F[2:7, 6:11] = 3.0                                          # lower lake
    F[7, 8] = 3.0
    F[8, 8] = 1.0                                               # outlet of the lower lake on the border
    ref, flat = masked_distances(F)
    got, _, nwalls = wall_distances(F)
    assert nwalls >= 1 and flat[4, 4] and flat[4, 6]
    np.testing.assert_array_equal(got[flat], ref[flat])
    assert ref[4, 4] == S                                       # fed by the spill cell; the lower lake is not (it drains through (7, 8))
    assert ref[4, 6] > S
