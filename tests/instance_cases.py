"""Parity cases that pin every kernel INSTANCE the two modexp launchers can select.

A launcher picks a template instance from (modulus bits, batch, groups, limbs per lane): the N^2
pair kernel ``powmod_n2_kernel<K, L, 29>`` (mx_powmod_nsquare_run) and the generic kernel
``powmod_kernel<K, L, 29, SLIDING>`` (mx_powmod_shared_lpl: one exponent, sliding window;
mx_powmod_multi_dev: per-group exponents, fixed window).  ``tests/test_instances.py`` (CPU) probes the
library's own geometry queries over the whole supported range and fails if an instance it can
return has no case below; ``tests/test_gpu_instances.py`` runs the cases bit-exactly against pow().

Each case: (kind, modulus-root bits, limbs_per_lane argument, batch, exponent bits[, wavefronts per group[, time-slice knob]]).
kind "n2": modulus is N^2 for an N of exactly that many bits; "shared" / "multi": modulus of that many bits.
The pair kernel exists in two forms — ``powmod_n2_kernel`` (one wavefront per group of elements) and
``powmod_n2_split_kernel`` (two: mx_powmod_n2_split.hpp) — selected by the sixth field (1 | 2; 0 = the
library's choice for this batch; 4 = the five-wavefront latency form ``powmod_n2_bipair_kernel``, mx_bipair.hpp); both have friendly-modulus instances that the library takes when the modulus leaves
the room (so the bit length of a case decides the instance), and the two-wavefront kernel has time-sliced instances
(seventh field 2: forced through the developer knob, as the automatic choice only takes them for batches of several
thousand).  An instance is the tuple ``mx_nsquare_launch_instance`` reports:
("n2", K, L, wavefronts per group, friendly, time-sliced).
"""

from __future__ import annotations

N2_CASES = [
    # one wavefront per group, narrow geometry (L = 9): K = 1, 2, 4, 8, 16, 16, 32, 32
    ("n2", 200, 9, 19, 130, 1), ("n2", 400, 9, 11, 130, 1), ("n2", 900, 9, 9, 130, 1), ("n2", 2051, 9, 9, 200, 1),
    ("n2", 3075, 9, 5, 96, 1), ("n2", 4099, 9, 5, 96, 1), ("n2", 6000, 9, 3, 64, 1), ("n2", 8200, 9, 3, 64, 1),
    # one wavefront per group, wide geometry (L = 18): K = 1, 2, 4, 8, 8, 16, 16
    ("n2", 400, 18, 70, 130, 1), ("n2", 900, 18, 40, 130, 1), ("n2", 2051, 18, 20, 200, 1),
    ("n2", 3075, 18, 12, 96, 1), ("n2", 4099, 18, 12, 96, 1), ("n2", 6000, 18, 6, 64, 1), ("n2", 8200, 18, 6, 64, 1),
    # two wavefronts per group, L = 9: K = 1, 2, 4, 8, 16, 16, 32, 32 (batches that leave the last workgroup ragged)
    ("n2", 200, 9, 130, 130, 2), ("n2", 400, 9, 67, 130, 2), ("n2", 900, 9, 35, 130, 2), ("n2", 2051, 9, 19, 200, 2),
    ("n2", 3075, 9, 9, 96, 2), ("n2", 4099, 9, 9, 96, 2), ("n2", 6000, 9, 5, 64, 2), ("n2", 8200, 9, 3, 64, 2),
    # two wavefronts per group, L = 18: K = 1, 2, 4, 8, 8, 16, 16
    ("n2", 400, 18, 70, 130, 2), ("n2", 900, 18, 40, 130, 2), ("n2", 2051, 18, 37, 200, 2),
    ("n2", 3075, 18, 12, 96, 2), ("n2", 4099, 18, 12, 96, 2), ("n2", 6000, 18, 6, 64, 2), ("n2", 8200, 18, 6, 64, 2),
    # two wavefronts per group, latency geometry L = 3: K = 1, 2, 4, 8, 16, 32, 64, 64
    ("n2", 50, 3, 130, 90, 2), ("n2", 130, 3, 67, 130, 2), ("n2", 300, 3, 35, 130, 2), ("n2", 600, 3, 19, 130, 2),
    ("n2", 1027, 3, 9, 130, 2), ("n2", 2051, 3, 5, 200, 2), ("n2", 3075, 3, 3, 96, 2), ("n2", 4099, 3, 3, 96, 2),
    # two wavefronts per group, L = 9, K = 8 and 16 with a modulus that leaves NO room for the friendly-modulus passes
    # (the cases at 2051 and 4099 bits above run the friendly instances, these the plain ones of the same geometry)
    ("n2", 2075, 9, 19, 200, 2), ("n2", 4160, 9, 9, 96, 2),
    # one wavefront per group, wide geometry, K = 4 and 8 with a modulus that leaves NO room for the friendly-modulus passes
    # (the cases at 2051 / 3075 / 4099 bits above run the friendly instances of round 4 — whose last segment is the plain
    # instance —, these the plain ones alone)
    ("n2", 2075, 18, 20, 200, 1), ("n2", 4160, 18, 12, 96, 1),
    # time-sliced instances of the two-wavefront kernel (L = 9, K = 1 .. 16; friendly and plain for K = 8 and 16)
    ("n2", 200, 9, 300, 130, 2, 2), ("n2", 400, 9, 67, 130, 2, 2), ("n2", 900, 9, 70, 130, 2, 2),
    ("n2", 2051, 9, 35, 200, 2, 2), ("n2", 2075, 9, 35, 200, 2, 2), ("n2", 4099, 9, 19, 96, 2, 2), ("n2", 4160, 9, 19, 96, 2, 2),
    # ... and at L = 18, K = 4 and 8 (round 5: what a lone launch just above one workgroup per CU runs, e.g. 10 000 ciphertexts
    # at key_length 2048); batches that leave the last group ragged, more groups than one workgroup's two pairs
    ("n2", 2051, 18, 70, 200, 2, 2), ("n2", 2075, 18, 37, 200, 2, 2), ("n2", 4099, 18, 35, 96, 2, 2),
    # FOUR wavefronts per group (round 6, csrc/mx_bipair.hpp: both passes of every pair product bipartite), latency geometry,
    # K = 16, 32 and 64 (key_length 1024 / 2048 / 4096); ragged last groups, a modulus that fills its geometry and one that does not
    ("n2", 1027, 3, 9, 130, 4), ("n2", 1100, 3, 5, 300, 4), ("n2", 2051, 3, 5, 200, 4), ("n2", 1700, 3, 7, 64, 4), ("n2", 2535, 3, 3, 96, 4),
    ("n2", 4099, 3, 3, 70, 4), ("n2", 3000, 3, 2, 64, 4), ("n2", 5300, 3, 1, 40, 4),
    # the library's choice: a handful of elements -> the latency geometry on four wavefronts where that form exists, on two elsewhere
    ("n2", 2051, 0, 7, 64, 0), ("n2", 4099, 0, 7, 64, 0),
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

# the latency instances of the generic kernel (3 limbs per lane, friendly-modulus products): K = 1 .. 64
GENERIC_CASES += [(kind, bits, 3, batch, ebits) for kind in ("shared", "multi") for bits, batch, ebits in
                  ((50, 70, 40), (130, 67, 100), (300, 35, 130), (600, 19, 130), (1027, 9, 130), (2051, 5, 130), (4100, 3, 96))]

# the bipartite latency form of the generic kernel (limbs_per_lane 6: 3 limbs per lane, every product on two wavefronts,
# csrc/mx_bimont.hpp): K = 4, 4, 8, 16, 16, 32, 64, 64 (the last at the widest modulus the form takes); fixed windows for
# one exponent and for per-group exponents alike; batches that leave the last workgroup ragged
GENERIC_CASES += [(kind, bits, 6, batch, ebits) for kind in ("shared", "multi") for bits, batch, ebits in
                  ((50, 70, 40), (130, 67, 100), (300, 35, 130), (600, 19, 130), (1027, 9, 130), (2051, 5, 130), (4100, 3, 96),
                   (5359, 2, 64))]

ALL_CASES = N2_CASES + GENERIC_CASES


def _geom(fn, *args):
    import ctypes

    k, l, w, b = (ctypes.c_int() for _ in range(4))
    rc = fn(*args, k, l, w, b)
    return (k.value, l.value) if rc == 0 else None


def _instance(lib, bits, batch, lpl, wpg, ts_knob=0):
    import ctypes

    k, l, wv, fr, ts = (ctypes.c_int() for _ in range(5))
    if ts_knob:
        assert lib.mx_debug_knob(3, ts_knob) == 0
    try:
        rc = lib.mx_nsquare_launch_instance(bits, batch, lpl, wpg, k, l, wv, fr, ts)
    finally:
        if ts_knob:
            lib.mx_debug_knob(3, 0)
    return (k.value, l.value, wv.value, fr.value, ts.value) if rc == 0 else None


def case_instance(lib, case):
    """The template instance a case runs: ("n2", K, L, wavefronts per group, friendly, time-sliced) or
    ("generic-sliding" | "generic-fixed", K, L) / ("generic-bi", K, 3)."""
    kind, bits, lpl, batch = case[:4]
    if kind == "n2":
        g = _instance(lib, bits, batch, lpl, case[5], case[6] if len(case) > 6 else 0)
        return None if g is None else ("n2",) + g
    groups = 1 if kind == "shared" else 3
    return _generic_instance(lib, bits, batch * groups, groups, lpl)


def _generic_instance(lib, bits, batch, groups, lpl):
    """("generic-sliding" | "generic-fixed", K, L) for the one-wavefront kernel, ("generic-bi", K, 3) for the bipartite form
    (mx_powmod_launch_form reports two wavefronts per group of elements)."""
    import ctypes

    g = _geom(lib.mx_powmod_geometry_for, bits, batch, groups, lpl)
    if g is None:
        return None
    waves, pivot = ctypes.c_int(), ctypes.c_int()
    assert lib.mx_powmod_launch_form(bits, batch, groups, lpl, waves, pivot) == 0
    if waves.value == 2:
        return ("generic-bi",) + g
    return ("generic-sliding" if groups == 1 else "generic-fixed",) + g


def reachable_instances(lib):
    """Every instance the launchers can return, probed through the library's own queries over the supported modulus
    range, a spread of batch sizes (among them the sizes just above a capacity step, where the time-sliced form is
    chosen) and all limbs_per_lane / wavefronts_per_group arguments."""
    out = set()
    bit_points = sorted(set(list(range(2, 600)) + list(range(600, 17000, 7)) + [16700, 16701, 8348, 8349, 4172, 4173]))
    batches = (1, 64, 2304, 4608, 5000, 5120, 9216, 10000, 18432, 20000, 36864, 40000, 73728, 80000, 147456, 160000, 200000, 2000000)
    for batch in batches:
        coarse = batch not in (1, 64, 5000, 20000, 200000, 2000000)
        for bits in (bit_points[::5] if coarse else bit_points):
            for lpl in (0, 3, 9, 18):
                for wpg in (0, 1, 2):
                    g = _instance(lib, bits, batch, lpl, wpg)
                    if g is not None:
                        out.add(("n2",) + g)
            if coarse:
                continue
            for lpl in (0, 3, 6, 9, 18):
                for groups in (1, max(2, batch // 40)):
                    g = _generic_instance(lib, bits, max(batch, groups), groups, lpl)
                    if g is not None:
                        out.add(g)
    return out
