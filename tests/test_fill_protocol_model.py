"""The worklist protocol of csrc/fill.hip (macro tiles of 2 x 2 tiles, sibling exchange through LDS, probes, marks) as a
deterministic CPU model (tools/fill_protocol_model.py), and the finding of round 3 it pins:

  rule "r02" (re-queue a sibling only when the exchange was capped) LOSES wake-ups -- a corner cell that drops in iteration 0
  of a visit whose iteration 1 is quiet never reaches the diagonal sibling; that is the one cell of 1.07 G cells left too high on
  a 16384 x 65536 raster in round 2;
  rule "r03" (a sibling's probe bit counts like any neighbour's), which fill.hip now implements, does not.

GPU: the raw schedule with its certification switched off (MHIP_FILL_NOCERTIFY) on random step terrains, which are full of
diagonal-only connections, against the oracle.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
import fill_protocol_model as model   # noqa: E402


def test_the_round_2_rule_loses_wake_ups_at_macro_tile_centres():
    bad, first = model.search("r02", trials=400, seed=5, shape=(8, 8), stop_at_first=True)
    assert bad == 1 and first is not None
    dem, got, truth = first
    rr, cc = np.nonzero(got != truth)
    assert np.all(got >= truth)
    # 8 x 8 cells, TI = 3: ONE macro tile; the cells left too high sit around its centre (rows / columns 3 and 4)
    assert set(rr.tolist()) <= {3, 4} and set(cc.tolist()) <= {3, 4, 5} | {2}


@pytest.mark.parametrize("shape,seed", [((14, 14), 1), ((17, 20), 2), ((8, 8), 5), ((26, 11), 3)])
def test_the_fixed_rule_reaches_the_fixed_point(shape, seed):
    bad, first = model.search("r03", trials=120, seed=seed, shape=shape)
    assert bad == 0, first


def test_the_model_restates_the_kernel_rule():
    """the rule the model calls r03 is the one in the kernel source (and the r02 rule is gone)"""
    src = (Path(__file__).resolve().parents[1] / "malstroem_amd" / "csrc" / "fill.hip").read_text()
    assert "const bool want = (((bits >> lane) & 1u) != 0) | (t == macro && capped);" in src
    assert "t == macro ? capped :" not in src


def diagonal_channel_dem(n_tiles=4, start=30):
    """A plateau at 5 with a one-cell channel at 0 along the main diagonal, open at the bottom-right corner only: the fill has
    to carry the level 0 up-left along the channel, across every macro-tile centre on the diagonal from a tile's corner cell
    (1, 1) to its diagonal sibling's corner cell (62, 62) -- the hand-over the round-2 rule dropped."""
    n = 2 + 62 * n_tiles
    dem = np.full((n, n), 5.0, dtype=np.float32)
    k = np.arange(start, n)
    dem[k, k] = 0.0
    return dem


@pytest.mark.gpu
def test_raw_iterative_schedule_carries_a_channel_across_macro_tile_centres(monkeypatch):
    """deterministic: with the round-2 rule the channel above the first macro-tile centre keeps the plateau level"""
    import oracle
    import malstroem_amd.algorithms as alg
    monkeypatch.setenv("MHIP_FILL", "iterative")
    monkeypatch.setenv("MHIP_FILL_NOCERTIFY", "1")
    for n_tiles, start in ((4, 30), (6, 5), (2, 10)):
        dem = diagonal_channel_dem(n_tiles, start)
        want = oracle.fill_terrain(dem)
        assert want[start + 1, start + 1] == 0.0        # the channel drains
        assert np.array_equal(alg.fill.fill_terrain(dem), want), (n_tiles, start)
        assert np.array_equal(alg.fill.fill_terrain(dem[::-1, ::-1].copy()), want[::-1, ::-1])   # ... and towards the other corner
        assert np.array_equal(alg.fill.fill_terrain(dem[::-1].copy()), want[::-1])              # ... and along the anti-diagonal


@pytest.mark.gpu
def test_raw_iterative_schedule_on_step_terrains(monkeypatch):
    import oracle
    import malstroem_amd.algorithms as alg
    monkeypatch.setenv("MHIP_FILL", "iterative")
    monkeypatch.setenv("MHIP_FILL_NOCERTIFY", "1")
    rng = np.random.default_rng(77)
    for k in range(24):
        h, w = int(rng.integers(130, 420)), int(rng.integers(130, 420))
        dem = rng.integers(0, 6, size=(h, w)).astype(np.float32)
        if k % 3 == 0:      # coarser plateaus
            dem = np.kron(rng.integers(0, 6, size=(h // 3 + 1, w // 3 + 1)), np.ones((3, 3)))[:h, :w].astype(np.float32)
        got = alg.fill.fill_terrain(dem)
        assert np.array_equal(got, oracle.fill_terrain(dem)), (k, h, w)
