"""GPU parity of every kernel instance the launchers can select (tests/instance_cases.py), bit-exact
against CPython pow(), plus the key_length-4096 (configs[4]) product path at the batch sizes of its
sweep with the automatically selected geometry asserted."""

from __future__ import annotations

import random

import pytest

import hostpow
import instance_cases as ic
from conftest import unhex
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from protocols.distributed_keygen_amd import Engine

    e = Engine()
    yield e
    e.set_limbs_per_lane(0)
    e.set_wavefronts_per_group(0)


def _modulus(rng, bits):
    return rng.getrandbits(bits) | (1 << (bits - 1)) | 1


@pytest.mark.parametrize("case", ic.ALL_CASES, ids=lambda c: f"{c[0]}-{c[1]}b-L{c[2]}-x{c[3]}" + (f"-w{c[5]}" if len(c) > 5 else "") + ("-ts" if len(c) > 6 else ""))
def test_instance_parity(eng, case):
    from protocols.distributed_keygen_amd import _lib

    kind, bits, lpl, batch, ebits = case[:5]
    rng = random.Random(hash(case) & 0xFFFFFF)
    eng.set_limbs_per_lane(lpl)
    eng.set_wavefronts_per_group(case[5] if kind == "n2" else 0)
    inst = ic.case_instance(_lib.lib(), case)
    assert inst is not None
    if kind == "n2":
        n = _modulus(rng, bits)
        n2 = n * n
        ts_knob = case[6] if len(case) > 6 else 0
        eng.debug_knob("n2_timeslice", ts_knob)
        try:
            shape = eng.nsquare_launch_shape(bits, batch)
            assert (shape[0], shape[1], shape[4]) == inst[1:4]
            assert bool(eng.nsquare_launch_timesliced(bits, batch)[0]) == bool(inst[5]) == bool(ts_knob)
            bases = [0, 1, n2 - 1, n, n + 1, n2 - n][: max(1, batch - 1)] + [rng.randrange(n2) for _ in range(batch)]
            bases = bases[:batch]
            for e in (rng.getrandbits(ebits) | (1 << (ebits - 1)) | 1, 2):
                assert eng.powmod_nsquare_batch(bases, e, n) == [pow(b, e, n2) for b in bases], (case, e.bit_length())
        finally:
            eng.debug_knob("n2_timeslice", 0)
    elif kind == "shared":
        mod = _modulus(rng, bits)
        assert eng.geometry(bits, batch, 1)[:2] == inst[1:]
        bases = ([0, 1, mod - 1] + [rng.randrange(mod) for _ in range(batch)])[:batch]
        e = rng.getrandbits(ebits) | (1 << (ebits - 1)) | 1
        assert eng.powmod_batch(bases, e, mod) == [pow(b, e, mod) for b in bases], case
    else:
        groups = 3
        mods = [_modulus(rng, bits - g) for g in range(groups)]
        assert eng.geometry(bits, batch * groups, groups)[:2] == inst[1:]
        exps = [rng.getrandbits(ebits - g) for g in range(groups)]
        rows = [([0, 1, m - 1] + [rng.randrange(m) for _ in range(batch)])[:batch] for m in mods]
        assert eng.powmod_batch_multi(rows, exps, mods) == [[pow(b, e, m) for b in r] for r, e, m in zip(rows, exps, mods)]
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)


def _special_moduli(bits):
    m = [(1 << bits) - 1, (1 << bits) - 3, (1 << (bits - 1)) + 1, (1 << bits) - (1 << (bits // 2)) - 1]
    return [x | 1 for x in m]


@pytest.mark.parametrize("bits", [3075, 4099, 6003])
def test_wide_pair_kernel_k8_k16_special_operands(eng, bits):
    """powmod_n2_kernel<8,18,29> / <16,18,29> (what key_length 3072 / 4096 launches from batch 3840)
    with the operands that stress the lazy-carry machinery: all-ones and sparse moduli, bases that
    are multiples of N, all-ones limb patterns in radix 2^29 and 2^32."""
    eng.set_limbs_per_lane(18)
    eng.set_wavefronts_per_group(1)
    rng = random.Random(bits)
    assert eng.nsquare_geometry(bits, 16)[:2] == ((8, 18) if bits <= 4172 else (16, 18))
    for n in _special_moduli(bits)[: 4 if bits < 5000 else 2]:
        n2 = n * n
        nb = n2.bit_length()
        pat29 = sum(((1 << 29) - 1) << (29 * k) for k in range(0, nb // 29 + 1, 2)) % n2
        pat32 = sum(0xFFFFFFFF << (32 * k) for k in range(0, nb // 32 + 1, 2)) % n2
        bases = [0, 1, n - 1, n, n + 1, 2 * n, n * (n - 1), n2 - 1, n2 - n, (n2 - 1) // 2, pat29, pat32,
                 (n2 - pat29) % n2, (n2 - pat32) % n2] + [rng.randrange(n2) for _ in range(4)]
        for e in (0, 1, 2, 3, (1 << 130) - 1, rng.getrandbits(150) | 1):
            assert eng.powmod_nsquare_batch(bases, e, n) == [pow(b, e, n2) for b in bases], (bits, e.bit_length())
            eng.set_wavefronts_per_group(2)        # and the two-wavefront form of the same instance
            assert eng.powmod_nsquare_batch(bases, e, n) == [pow(b, e, n2) for b in bases], (bits, e.bit_length(), "split")
            eng.set_wavefronts_per_group(1)
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)


@pytest.mark.parametrize("lpl,wpg", [(9, 1), (18, 1), (3, 2), (9, 2), (18, 2)])
def test_golden_key4096_partial_decryptions_every_launch_shape(eng, golden_decrypt_synth, lpl, wpg):
    """The reference-generated partial decryptions of the key_length-4096 key through every launch shape of the
    pair kernel: narrow <16,9> and wide <8,18> on one wavefront per group, and <64,3>, <16,9>, <8,18> on two."""
    grp = golden_decrypt_synth["k4096_n3_t1"]
    n = unhex(grp["n"])
    n2 = n * n
    eng.set_limbs_per_lane(lpl)
    eng.set_wavefronts_per_group(wpg)
    assert eng.nsquare_geometry(n.bit_length(), 3)[:2] == {9: (16, 9), 18: (8, 18), 3: (64, 3)}[lpl]
    cs = [unhex(c["c"]) for c in grp["cases"]]
    for i, share in grp["shares"].items():
        exp = oracle.partial_decrypt_exponent(int(i), grp["degree"], unhex(grp["n_fac"]), unhex(share))
        bases = cs if exp >= 0 else [oracle.mod_inv(c, n2) for c in cs]
        assert eng.powmod_nsquare_batch(bases, abs(exp), n) == [unhex(c["partials"][i]) for c in grp["cases"]], i
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)


@pytest.mark.parametrize("batch,shape", [(1024, (64, 3, 2)), (4096, (8, 18, 2)), (16384, (8, 18, 1))])
def test_c5_sweep_points_auto_geometry(eng, batch, shape):
    """configs[4] at the three batch sizes of its sweep (1024 / 4096 / 16 384 ciphertexts at key_length 4096) through the
    product path with the library's own choice of launch shape — a different instance at every size: the latency
    geometry on two wavefronts, the wide <8,18> instance on two wavefronts, the wide one-wavefront instance (on the
    friendly modulus) —: 64-96 samples bit-exact against pow() on the host cores, and the full threshold decryption round
    trip decrypt(encrypt(m)) == m on every ciphertext."""
    from protocols.distributed_keygen_amd import synthetic

    key = synthetic.make_key(4096, 3, 1)
    n, n2 = key.n, key.n_square
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)
    got_shape = eng.nsquare_launch_shape(n.bit_length(), batch)
    assert got_shape[:2] + got_shape[4:] == shape
    assert eng.nsquare_launch_timesliced(n.bit_length(), batch) == (0, 0)
    rng = random.Random(40960 + batch)
    msgs = [rng.randrange(n) for _ in range(batch)]
    msgs[:3] = [0, 1, n - 1]
    rs = [rng.randrange(1, n) for _ in range(batch)]
    cts = eng.encrypt_batch(msgs, rs, n)                               # r^N through the same kernel
    assert cts[5] == (1 + msgs[5] * n) * pow(rs[5], n, n2) % n2
    partials = []
    for i in range(1, key.degree + 2):
        e = key.exponent(i)
        bases = cts if e >= 0 else eng.modinv_batch(cts, n2)
        partials.append(eng.powmod_nsquare_batch(bases, abs(e), n))
    got, ok = eng.combine_batch([[partials[i][k] for i in range(key.degree + 1)] for k in range(batch)], n, key.theta_inv)
    assert all(ok) and got == msgs
    i_pos = next(i for i in (1, 2, 3) if key.exponent(i) >= 0)
    idx = [0, 1, 2, batch - 1] + [(k * 7919) % batch for k in range(1, 93 if batch == 4096 else 61)]
    want = hostpow.powmod_many([(cts[k], key.exponent(i_pos), n2) for k in idx], chunksize=1)
    assert [partials[i_pos - 1][k] for k in idx] == want
    if batch == 4096:
        # the one-wavefront narrow and wide instances on the same inputs agree on every ciphertext
        for lpl, want_geo in ((9, (16, 9)), (18, (8, 18))):
            eng.set_limbs_per_lane(lpl)
            eng.set_wavefronts_per_group(1)
            assert eng.nsquare_geometry(n.bit_length(), batch)[:2] == want_geo
            assert eng.powmod_nsquare_batch(cts, key.exponent(i_pos), n) == partials[i_pos - 1]
        # and the plain one-wavefront wide instance (friendly-modulus passes switched off) agrees with the friendly one
        eng.debug_knob("n2_friendly_1w", 1)
        try:
            assert eng.powmod_nsquare_batch(cts, key.exponent(i_pos), n) == partials[i_pos - 1]
        finally:
            eng.debug_knob("n2_friendly_1w", 0)
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)
