"""CPU-only: the decomposition behind the accumulation's final pass (accum.hip: accum_final_walk_kernel).

final[c] = tile-local sum[c] + the external inflow of every ENTRY cell (a cell with an upstream neighbour outside its 64 x 64 tile)
whose path inside the tile runs through c.  Stated in numpy with the oracle as the tile-local accumulation, checked against the oracle's
accumulation of the whole raster; and the numbers the kernel's design quotes (entries per tile, length of their paths)."""
import numpy as np

import oracle
from _cases import fbm

T = 64
DR = np.array([-1, -1, 0, 1, 1, 1, 0, -1])
DC = np.array([0, 1, 1, 1, 0, -1, -1, -1])


def test_local_sums_plus_entry_walks_give_the_accumulation():
    dem = fbm(200, 330, seed=31)           # ragged tiles on both axes
    s, d = oracle.minimum_safe_short_and_diag(dem)
    fd = oracle.terrain_flowdirection(oracle.fill_terrain_no_flats(dem, s, d))
    want = oracle.accumulated_flow(fd)
    h, w = fd.shape
    got = np.zeros_like(want)
    steps = entries = 0
    for r0 in range(0, h, T):
        for c0 in range(0, w, T):
            tile = fd[r0:r0 + T, c0:c0 + T]
            th, tw = tile.shape
            local = oracle.accumulated_flow(np.ascontiguousarray(tile))      # what leaves the tile leaves, nothing comes in
            # the inflow of the entry cells: the final values of the upstream neighbours outside the tile
            for r in range(th):
                for c in range(tw):
                    if 0 < r < th - 1 and 0 < c < tw - 1:
                        continue                       # (entries lie on the tile's outline)
                    inflow = 0.0
                    for k in range(8):
                        ur, uc = r0 + r + DR[k], c0 + c + DC[k]      # neighbour k; it flows into (r, c) iff its code is the opposite direction
                        if not (0 <= ur < h and 0 <= uc < w) or (r0 <= ur < r0 + th and c0 <= uc < c0 + tw):
                            continue
                        if fd[ur, uc] == (k + 4) % 8:
                            inflow += want[ur, uc]
                    if inflow == 0:
                        continue
                    entries += 1
                    pr, pc = r, c                      # the walk: add the inflow along the tile-local path
                    while True:
                        local[pr, pc] += inflow
                        steps += 1
                        code = tile[pr, pc]
                        if code > 7:
                            break
                        pr, pc = pr + DR[code], pc + DC[code]
                        if not (0 <= pr < th and 0 <= pc < tw):
                            break
            got[r0:r0 + th, c0:c0 + tw] = local
    assert np.array_equal(got, want)
    ntiles = -(-h // T) * -(-w // T)
    assert entries / ntiles < 252 / 2 and steps / max(entries, 1) < 64      # few entries, short paths: what makes walking cheaper than doubling
