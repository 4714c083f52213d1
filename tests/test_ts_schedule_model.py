"""The discrete-event model behind the unit scheduler of the time-sliced pair-kernel launch (tools/ts_schedule_model.py,
csrc/mx_powmod_n2_split.hpp): what the two disciplines cost for the shapes the library takes, and the library's own choice
of units per group against the model's rounds."""
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tools"))

import ts_schedule_model as model  # noqa: E402


def test_fifo_queue_loses_a_round_that_most_work_left_first_does_not():
    # 10 000 ciphertexts at key_length 2048: 625 groups on 512 resident pairs, a full launch of <= 512 groups = 32 ms
    fluid = 32.0 * 625 / 512
    assert model.simulate(625, 512, 4, policy="fifo") > fluid * 1.18          # measured 50.7-51.4 ms (rounds 4, 5)
    assert model.simulate(625, 512, 4, policy="lrf") < fluid * 1.05
    for units in (8, 12, 16):
        assert model.simulate(625, 512, units, policy="lrf") <= model.simulate(625, 512, units, policy="fifo") + 0.1
        assert model.simulate(625, 512, units, policy="lrf") < fluid * 1.04
    # with no more groups than pairs there is nothing to schedule
    assert abs(model.simulate(512, 512, 8, policy="lrf") - 32.0) < 1.0 and abs(model.simulate(512, 512, 8, policy="fifo") - 32.0) < 1.0


def test_units_per_group_the_library_picks_are_whole_rounds_of_the_model():
    from protocols.distributed_keygen_amd import _lib

    lib = _lib.lib()
    pairs = 512            # one workgroup of two pairs on each of 256 CUs (the library asks the device; without one it assumes 256)
    for batch in (8448, 8704, 9216, 10000, 10240, 10752, 11264, 12288):
        r, u = ctypes.c_int(), ctypes.c_int()
        assert lib.mx_nsquare_launch_timesliced(2051, batch, 0, 0, r, u) == 0
        assert r.value == 1 and u.value in (2, 8, 12), (batch, r.value, u.value)
        groups = -(-batch // 16)
        cost = lambda units: -(-groups * units // pairs) / units * (1 + 0.004 * units)
        assert cost(u.value) == min(cost(x) for x in (2, 8, 12)), batch
        # and the model agrees that this many units come within a few per cent of the rounds counted
        t = model.simulate(groups, pairs, u.value, policy="lrf")
        assert t <= 32.0 * (-(-groups * u.value // pairs)) / u.value * 1.03 + 0.5, (batch, u.value, t)
