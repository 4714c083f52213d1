"""Bounded slices of the randomised soaks (tools/soak_round5.py, tools/soak_timesliced.py) inside `pytest -m gpu`: fixed
seeds — among them the seeds that found the bipartite form's off-by-N (19) and ran on the hand-over fix (11) — and a fixed
number of rounds, so that the differential machinery that found both of round 5's wrong-result bugs runs on every library
the suite runs on.  The open-ended runs stay in tools/ (records in profiles/r0N_soak.txt)."""

from __future__ import annotations

import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))


@pytest.fixture(scope="module")
def eng():
    from protocols.distributed_keygen_amd import Engine

    e = Engine()
    yield e
    e.debug_knob("n2_timeslice", 0)
    e.set_limbs_per_lane(0)
    e.set_wavefronts_per_group(0)
    e.set_segments(0)


@pytest.mark.parametrize("seed,rounds", [(19, 45), (5, 25), (31, 25)])
def test_round5_soak_slice(eng, seed, rounds):
    """Bipartite latency form (random lengths, mixed-length groups, pivots, lanes) and the fixed-window tape in random
    launch shapes, every row against CPython pow."""
    import soak_round5

    done_rounds, counts = soak_round5.soak(eng, seed, rounds_limit=rounds)
    assert done_rounds == rounds and counts["bipartite"] > 0 and counts["fixed_window"] > 0


@pytest.mark.parametrize("seed,launches", [(11, 30), (23, 20)])
def test_timesliced_soak_slice(eng, seed, launches):
    """Time-sliced launches over random key lengths, batches around the resident capacity, 1..16 units, 1..3 workgroups
    per CU, alone or two at once on two streams: every row equal to the plain one-wavefront launch's."""
    import soak_timesliced

    done, rows, by_shape = soak_timesliced.soak(eng, seed, launches_limit=launches)
    assert done >= launches and rows > 0
    assert any(form == "sliced" for (_l, _k, form) in by_shape), by_shape


@pytest.mark.parametrize("seed,rounds", [(3, 40), (29, 40)])
def test_bipair_soak_slice(eng, seed, rounds):
    """The five-wavefront latency form of the pair kernel (csrc/mx_bipair.hpp) over random modulus lengths of its whole range,
    special moduli and bases, exponents of 1 bit .. full length, sliding and fixed-window tapes, batches of 1 .. 700, now and
    then two launches at once: every result against CPython pow."""
    import soak_bipair

    done_rounds, modexps = soak_bipair.soak(eng, seed, rounds_limit=rounds)
    assert done_rounds == rounds and modexps > 0
