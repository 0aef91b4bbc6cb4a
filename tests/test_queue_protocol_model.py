"""The flood's queue-driven seed-graph solve (csrc/pflood.hip: pf_solve_queue_body -- block visits taken from one queue by a resident
grid, round 4) as a CPU model under random interleavings (tools/queue_protocol_model.py), and the three findings it pins:

  "r04"       what the kernel does -- levels lowered by atomic MINs, ONE visit of a block at a time (mark word: queued | running; a block
              woken while it runs is queued again when its visit ends), end of the solve = finished == tail -- reaches the exact minimax
              levels on every schedule tried and never goes below them;
  "first"     the first version of the round (plain stores, a woken block queued at once even while a visit of it runs): two visits of
              one block overlap and the one that started from the older levels puts a seed back UP -- on the GPU the flood's run-time
              proof failed in 4 of 8 steps;
  "late_old"  the block's levels "as loaded" copied behind the first relaxations (a pass of its own behind the barrier the sweeps start
              at -- in the rounds' kernel since round 2): a drop that a fast thread made before a slow thread's copy is never written
              back and nobody is woken -- seen on the GPU when a 64-register build skewed the waves.
"""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
import queue_protocol_model as model   # noqa: E402


@pytest.mark.parametrize("nbr,nbc,workers,seed", [(4, 4, 4, 1), (3, 6, 3, 2), (5, 5, 8, 3), (2, 2, 2, 4), (1, 7, 3, 5)])
def test_the_kernels_protocol_is_exact_on_every_schedule(nbr, nbc, workers, seed):
    bad, first, below = model.search("r04", trials=150, seed=seed, nbr=nbr, nbc=nbc, nworkers=workers)
    assert bad == 0 and not below, first


def test_overlapping_visits_of_one_block_put_a_seed_back_up():
    bad, first, below = model.search("first", trials=300, seed=1)
    assert bad > 0 and not below
    got, want = first
    assert all(g >= w for g, w in zip(got, want)) and got != want        # an upper bound that is not the solution: what check.hip catches


def test_levels_copied_behind_the_first_relaxations_lose_drops():
    bad, first, below = model.search("late_old", trials=300, seed=1)
    assert bad > 0 and not below
    got, want = first
    assert all(g >= w for g, w in zip(got, want)) and got != want


def test_the_model_restates_the_kernel():
    """the pieces the model calls "r04" are in the kernel source: atomic MIN write-back, the running bit, the copies with the loads"""
    src = (Path(__file__).resolve().parents[1] / "malstroem_amd" / "csrc" / "pflood.hip").read_text()
    body = src[src.index("void pf_solve_queue_body("):src.index("void pf_solve_queue_kernel(")]
    assert "atomicMin(&a.Lv[" in body and "st_sc1(&a.Lv[" not in body
    assert "atomicExch(&sa.mark_cur[blk], 2u)" in body and "atomicOr(&sa.mark_cur[nb], 1u)" in body and "atomicAnd(&sa.mark_cur[blk], ~2u)" in body
    loads = body[body.index("levels of the region"):body.index("the block's relaxations (read-only")]
    assert "Lold[" in loads                                                # written with the loads, in front of the barrier
    sweeps = body[body.index("the block's fixed point"):body.index("write back")]
    assert "Lold[" not in sweeps
