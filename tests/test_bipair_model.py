"""CPU: the column-exact model of the BIPARTITE PAIR product (tools/bipair_model.py — the arithmetic of csrc/mx_bipair.hpp,
the five-wavefront latency form of the N^2 pair kernel) against big-integer arithmetic modulo N^2.  The model asserts every
width the kernel relies on while it runs: 64-bit lazy columns in all four wavefronts, the 32-bit words that cross lanes
(29-bit limb + carry word in pass 1, 30-bit limb + carry word of weight 2 in pass 2), the fold digits, the quotient
columns c * Vq + sum dg_k cf_k; tools/bipair_debug.py compares the kernel's slots with this model limb for limb on the GPU."""

from __future__ import annotations

import random
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
import bipair_model as bp  # noqa: E402
from bimont_model import Geometry, L, MASK, W, limbs_of, value_of  # noqa: E402


@pytest.mark.parametrize("bits", [1027, 1029, 2051, 2053, 1500, 4099])
def test_chains_of_pair_products_match_big_integer_arithmetic(bits):
    rng = random.Random(bits)
    for n in (rng.getrandbits(bits) | (1 << (bits - 1)) | 1, (1 << bits) - 1, (1 << (bits - 1)) + 1):
        geo = bp.pair_geometry(bits)
        cst = bp.PairConstants(n, geo)
        n2 = n * n
        track = {}
        x = rng.randrange(n2)
        X = bp.to_pair(geo, cst, x)
        acc, val = X, x
        for step in range(7 if bits < 3000 else 4):          # (a pure-Python product of 147 limb steps on 64 lanes takes seconds)
            if step % 3 == 2:
                acc, val = bp.pair_mul(geo, cst, acc, X, False, track), val * x % n2
            else:
                acc, val = bp.pair_mul(geo, cst, acc, acc, True, track), val * val % n2
            assert bp.pair_value(cst, acc) == val, (bits, step)
        assert track["v"] < (1 << W) + 64 and track["cy"] < (1 << 32) and track["v2"] < (1 << (W + 1)) + 64 and track["cy2"] < (1 << 32)
        assert track["top"] <= 4 and track["top2"] <= 4 and track["c_limbs"] <= 11


def test_every_width_holds_for_operands_at_their_bounds():
    """Operands whose limbs all sit at the lazy bound 2^29 + 2^7 - 1 (and limb Pd at 2), for both digits of both factors:
    the value identity and every asserted width (fold digits, carry words, 64-bit columns, quotient columns) still hold."""
    rng = random.Random(5)
    bits = 2053
    geo = bp.pair_geometry(bits)
    cnt = L * geo.K
    big = (1 << W) + (1 << 7) - 1

    def mk(kind):
        v = [0] * cnt
        for i in range(geo.Pd):
            v[i] = big if kind == "max" else (MASK if kind == "ones" else rng.randrange(big + 1))
        v[geo.Pd] = 2 if kind == "max" else 0
        return v

    for n in ((1 << bits) - 1, (1 << (bits - 1)) + 1, rng.getrandbits(bits) | (1 << (bits - 1)) | 1):
        cst = bp.PairConstants(n, geo)
        n2 = n * n
        for kx, ky in (("max", "max"), ("max", "ones"), ("ones", "rand")):
            X, Y = (mk(kx), mk(kx)), (mk(ky), mk(ky))
            vx, vy = bp.pair_value(cst, X), bp.pair_value(cst, Y)
            assert bp.pair_value(cst, bp.pair_mul(geo, cst, X, Y, False)) == vx * vy % n2
            assert bp.pair_value(cst, bp.pair_mul(geo, cst, X, X, True)) == vx * vx % n2


def test_the_whole_flow_with_the_kernels_constants():
    """Conversion into the domain by two pair products with K1 = digits(2^(2 W hL)), K2 = digits(2^(2 W hL + k)) of the same
    n2_constants the two-wavefront kernel uses (for R = 2^(W hL) instead of 2^(W Pd)), and the way out: ONE plain pair product
    (radix 2^(W Pd), as the two-wavefront kernel's last segment runs it) by E = (2^(W (Pd - hL)), 0) leaves digits below 2 N
    whose N-adic value is the residue."""
    rng = random.Random(9)
    for bits in (2053, 1029, 4099):
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        n2 = n * n
        geo = bp.pair_geometry(bits)
        cst = bp.PairConstants(n, geo)
        cnt = L * geo.K
        m, k = W * geo.h_lo, bits - 1
        pair_of = lambda v: (limbs_of(v % n2 % n, cnt), limbs_of(v % n2 // n, cnt))
        x = rng.randrange(n2)
        xlo, xhi = x & ((1 << k) - 1), x >> k
        a = bp.pair_mul(geo, cst, (limbs_of(xlo, cnt), [0] * cnt), pair_of(1 << (2 * m)))
        b = bp.pair_mul(geo, cst, (limbs_of(xhi, cnt), [0] * cnt), pair_of(1 << (2 * m + k)))
        s = ([p + q for p, q in zip(a[0], b[0])], [p + q for p, q in zip(a[1], b[1])])
        assert bp.pair_value(cst, s) == x
        r = 1 << (W * geo.Pd)
        x0, x1, e0 = value_of(s[0]), value_of(s[1]), 1 << (W * (geo.Pd - geo.h_lo))
        assert e0 < n

        def redc(t):
            q = (-t * pow(n, -1, r)) % r
            return (t + q * n) // r, q

        z0, q = redc(x0 * e0)
        z1, _ = redc(x1 * e0 + n * (-(-r // n)) - q)
        assert z0 < 2 * n and z1 < 2 * n + 2 and (z0 + z1 * n) % n2 == x


def test_the_librarys_geometry_of_the_form_is_the_models():
    """mx_nsquare_latency_form (ABI 4.4): lanes, data positions and the pair form's own pivot as tools/bipair_model.py's
    pair_geometry has them, over the three ranges of instances; refused below, between and beyond them."""
    import ctypes

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from protocols.distributed_keygen_amd import _lib

    lib = _lib.lib()
    for bits in (830, 1027, 1029, 1500, 2051, 2053, 2535, 2900, 3100, 4099, 4102, 5300):
        geo = bp.pair_geometry(bits)
        k, pd, pivot, most = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        assert lib.mx_nsquare_latency_form(bits, k, pd, pivot, most) == 0, bits
        assert (k.value, pd.value, pivot.value) == (geo.K, geo.Pd, geo.h_lo), (bits, k.value, pd.value, pivot.value, geo.K, geo.Pd, geo.h_lo)
        assert most.value % (64 // geo.K) == 0 and most.value > 0
    assert {bits: bp.pair_geometry(bits).h_lo for bits in (1027, 2051, 4099)} == {1027: 21, 2051: 39, 4099: 75}
    for bits in (300, 2600, 5700):
        assert lib.mx_nsquare_latency_form(bits, None, None, None, None) == -2      # MX_ERR_SIZE
