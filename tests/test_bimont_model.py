"""CPU: the column-exact model of the bipartite latency modexp (tools/bimont_model.py — the arithmetic of
csrc/mx_bimont.hpp with the kernel's lazy 64-bit columns, 29-bit limbs and neighbour exchanges) against pow().
The model asserts every width the kernel relies on while it runs: 64-bit columns, 32-bit words crossing lanes, the
32-bit fold digit.  The geometry must be the one the library launches (mx_powmod_launch_form / mx_powmod_geometry_for)."""

from __future__ import annotations

import ctypes
import random
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
import bimont_model as bm  # noqa: E402


@pytest.mark.parametrize("bits", [3, 50, 131, 515, 1026, 1029, 2050, 2053, 3000, 5359])
def test_model_matches_pow(bits):
    rng = random.Random(bits)
    track = {}
    specials = [(1 << bits) - 1, (1 << (bits - 1)) + 1, ((1 << bits) - 1) ^ (1 << (bits // 2))]
    for trial in range(3 if bits < 3000 else 1):
        n = (specials[trial] if trial < 2 and bits > 3 else rng.getrandbits(bits) | (1 << (bits - 1))) | 1
        for g in (rng.randrange(n), n - 1, 0):
            e = rng.choice([rng.getrandbits(24) | 1, (1 << 20) - 1, 1, 0])
            assert bm.powmod(g, e, n, win=rng.choice([1, 2, 5]), track=track) == pow(g, e, n)
    assert track.get("v", 0) < (1 << 30) and track.get("cy", 0) < (1 << 32) and track.get("fin", 0) < (1 << 30)


def test_every_pivot_gives_the_same_product():
    """The pivot only moves limb steps between the two wavefronts: any multiple of 3 inside the multiplier works."""
    rng = random.Random(7)
    n = rng.getrandbits(300) | (1 << 299) | 1
    g, e = rng.randrange(n), rng.getrandbits(40)
    geo = bm.Geometry(300)
    for h_lo in range(3, geo.Pd + 3, 3):
        assert bm.powmod(g, e, n, h_lo=h_lo) == pow(g, e, n), h_lo


def test_library_launches_the_models_geometry():
    from protocols.distributed_keygen_amd import _lib

    lib = _lib.lib()
    for bits in (3, 50, 58, 87, 88, 131, 300, 600, 1026, 1029, 2050, 2053, 4100, 5359):
        geo = bm.Geometry(bits)
        k, l, w, b, waves, pivot = (ctypes.c_int() for _ in range(6))
        assert lib.mx_powmod_geometry_for(bits, 80, 2, 6, k, l, w, b) == 0
        assert lib.mx_powmod_launch_form(bits, 80, 2, 6, waves, pivot) == 0
        assert (k.value, l.value, w.value, 3 * b.value, waves.value, pivot.value) == (geo.K, 3, 29, geo.Pd, 2, geo.h_lo), bits
    k, l, w, b = (ctypes.c_int() for _ in range(4))
    assert lib.mx_powmod_geometry_for(5360, 80, 2, 6, k, l, w, b) == -2          # MX_ERR_SIZE: the form ends at 5359 bits
    waves = ctypes.c_int()
    assert lib.mx_powmod_launch_form(2053, 80, 2, 9, waves, None) == 0 and waves.value == 1


def test_leaving_the_domain_needs_the_factor_reduced_modulo_each_groups_modulus():
    """The epilogue (one plain Montgomery product with theta * R, two conditional subtractions) for the largest lazy
    accumulator a product can leave, with the launch's geometry taken from a LONGER modulus than the group's own: correct
    with the factor reduced modulo the group's N (bisetup_kernel, row 8), and off by a multiple of N with the bare power
    of two the first form of the kernel used (tools/soak_round5.py seed 19; tests/test_gpu_powmod.py has the GPU case)."""
    import random

    rng = random.Random(19)
    broken = 0
    for launch_bits in (91, 1029, 2053):
        for shorter in (0, 1, 5, 17, launch_bits // 2):
            nbits = launch_bits - shorter
            n = rng.getrandbits(nbits) | (1 << (nbits - 1)) | 1
            for h_lo in (3, 0):
                geo = bm.Geometry(launch_bits, h_lo)
                theta = pow(pow(2, bm.W * geo.h_lo, n), -1, n)
                top = (1 << (bm.W * geo.Pd)) + (1 << (bm.W * geo.Pd - 20))          # < 2^(W*Pd) * (1 + tiny)
                for r in (1, n - 1, rng.randrange(n)):
                    acc = r + (top - r) // n * n                                      # the largest representative of r
                    assert bm.leave_domain(acc, n, geo, reduced=True) == r * theta % n, (launch_bits, shorter, h_lo)
                    broken += bm.leave_domain(acc, n, geo, reduced=False) != r * theta % n
    assert broken > 0          # (the bare power of two is only safe for moduli of the launch's own length)
