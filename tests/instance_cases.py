"""Parity cases that pin every kernel INSTANCE the two modexp launchers can select.

A launcher picks a template instance from (modulus bits, batch, groups, limbs per lane): the N^2
pair kernel ``powmod_n2_kernel<K, L, 29>`` (mx_powmod_nsquare_run) and the generic kernel
``powmod_kernel<K, L, 29, SLIDING>`` (mx_powmod_shared_lpl: one exponent, sliding window;
mx_powmod_multi_dev: per-group exponents, fixed window).  ``tests/test_instances.py`` (CPU) probes the
library's own geometry queries over the whole supported range and fails if an instance it can
return has no case below; ``tests/test_gpu_instances.py`` runs the cases bit-exactly against pow().

Each case: (kind, modulus-root bits, limbs_per_lane argument, batch, exponent bits).
kind "n2": modulus is N^2 for an N of that many bits; "shared" / "multi": modulus of that many bits.
"""

from __future__ import annotations

N2_CASES = [
    # narrow geometry (L = 9): K = 1, 2, 4, 8, 16, 16, 32, 32
    ("n2", 200, 9, 19, 130), ("n2", 400, 9, 11, 130), ("n2", 900, 9, 9, 130), ("n2", 2051, 9, 9, 200),
    ("n2", 3075, 9, 5, 96), ("n2", 4099, 9, 5, 96), ("n2", 6000, 9, 3, 64), ("n2", 8200, 9, 3, 64),
    # wide geometry (L = 18): K = 1, 2, 4, 8, 8, 16, 16
    ("n2", 400, 18, 70, 130), ("n2", 900, 18, 40, 130), ("n2", 2051, 18, 20, 200),
    ("n2", 3075, 18, 12, 96), ("n2", 4099, 18, 12, 96), ("n2", 6000, 18, 6, 64), ("n2", 8200, 18, 6, 64),
    # automatic choice: small batch -> narrow
    ("n2", 2051, 0, 7, 64),
]

GENERIC_CASES = [
    # one exponent (sliding window), narrow: K = 1 .. 64
    ("shared", 200, 9, 9, 130), ("shared", 400, 9, 9, 130), ("shared", 900, 9, 9, 130), ("shared", 2051, 9, 9, 130),
    ("shared", 4100, 9, 5, 96), ("shared", 8200, 9, 3, 64), ("shared", 16400, 9, 2, 40),
    # one exponent, wide: K = 1 .. 32
    ("shared", 400, 18, 70, 130), ("shared", 900, 18, 40, 130), ("shared", 2051, 18, 20, 130),
    ("shared", 4100, 18, 10, 96), ("shared", 8200, 18, 6, 64), ("shared", 16400, 18, 3, 40),
    # per-group exponents (fixed window), narrow and wide
    ("multi", 200, 9, 9, 130), ("multi", 400, 9, 9, 130), ("multi", 900, 9, 9, 130), ("multi", 2051, 9, 9, 130),
    ("multi", 4100, 9, 5, 96), ("multi", 8200, 9, 3, 64), ("multi", 16400, 9, 2, 40),
    ("multi", 400, 18, 70, 130), ("multi", 900, 18, 40, 130), ("multi", 2051, 18, 20, 130),
    ("multi", 4100, 18, 10, 96), ("multi", 8200, 18, 6, 64), ("multi", 16400, 18, 3, 40),
]

ALL_CASES = N2_CASES + GENERIC_CASES


def _geom(fn, *args):
    import ctypes

    k, l, w, b = (ctypes.c_int() for _ in range(4))
    rc = fn(*args, k, l, w, b)
    return (k.value, l.value) if rc == 0 else None


def case_instance(lib, case):
    """The template instance ("n2" | "generic-sliding" | "generic-fixed", K, L) a case runs."""
    kind, bits, lpl, batch, _ = case
    if kind == "n2":
        g = _geom(lib.mx_nsquare_geometry_for, bits, batch, lpl)
        return None if g is None else ("n2",) + g
    groups = 1 if kind == "shared" else 3
    g = _geom(lib.mx_powmod_geometry_for, bits, batch * groups, groups, lpl)
    return None if g is None else ("generic-sliding" if kind == "shared" else "generic-fixed",) + g


def reachable_instances(lib):
    """Every instance the launchers can return, probed through the library's geometry queries over the
    supported modulus range, a spread of batch sizes and all three limbs_per_lane arguments."""
    out = set()
    bit_points = sorted(set(list(range(2, 600)) + list(range(600, 17000, 7)) + [16700, 16701, 8348, 8349, 4172, 4173]))
    for lpl in (0, 9, 18):
        for batch in (1, 64, 5000, 20000, 200000, 2000000):
            for bits in bit_points:
                g = _geom(lib.mx_nsquare_geometry_for, bits, batch, lpl)
                if g is not None:
                    out.add(("n2",) + g)
                for groups, name in ((1, "generic-sliding"), (max(2, batch // 40), "generic-fixed")):
                    g = _geom(lib.mx_powmod_geometry_for, bits, max(batch, groups), groups, lpl)
                    if g is not None:
                        out.add((name,) + g)
    return out
