"""GPU: patch.install() closures on the HIP engine (VERDICT r02 "next" 1b).  The stand-in package
(tests/standin: the reference's class surface, builder-written) is patched with the REAL engine and its
coroutines are driven by three in-process parties: `_decrypt_sequence_raw` with the party's own column on the
device and the peers' columns arriving as plain ints / wire-form integers / non-canonical residues, the single
`_decrypt_raw`, the reference's error behaviour, `compute_modulus` end to end (reconstruct -> sieve -> Jacobi ->
modexps -> verdict, one launch each per round), and the rebound arithmetic leaf."""

from __future__ import annotations

import random

import pytest
import sympy

import standin_harness as sh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from protocols.distributed_keygen_amd import Engine

    return Engine()


@pytest.mark.parametrize("key_length,count", [(1024, 48), (2048, 6)])
def test_patched_decrypt_sequence_on_the_hip_engine(eng, key_length, count):
    from protocols.distributed_keygen_amd import codec, patch, synthetic

    key = synthetic.make_key(key_length, 3, 1)
    rng = random.Random(key_length)
    msgs = [0, 1, key.n - 1] + [rng.randrange(key.n) for _ in range(count - 3)]
    cts = [synthetic.encrypt(key, m, rng) for m in msgs]
    n2 = key.n_square

    def wire(sender, vals):
        if sender == "p2":                         # as a real transport delivers big integers
            return [codec.encode_int(v) for v in vals]
        if sender == "p3":                         # a peer that does not canonicalise: same residues, other representatives
            return [v + n2 if k % 2 else codec.encode_int(v + 3 * n2) for k, v in enumerate(vals)]
        return vals

    patch.install(engine=eng, package=sh.PACKAGE)
    try:
        parties = sh.parties_for_key(key, wire=wire)
        cobjs = sh.ciphertexts(key, cts)
        got = sh.decrypt_sequence(parties, cobjs)
        assert [[e.value for e in r] for r in got] == [msgs] * 3
        assert all(type(e.value) is int for r in got for e in r) and all(not c.fresh for c in cobjs)
        # the partial decryptions that went over the "network" are the reference's (CPython pow on a sample)
        own = parties[0].secret_key.partial_decrypt(sh.ciphertexts(key, cts[:1])[0])
        e1 = key.exponent(1)
        assert own == (pow(cts[0], e1, n2) if e1 >= 0 else pow(pow(cts[0], -1, n2), -e1, n2))
        # single decrypt (DK:314-382 shape)
        assert [e.value for e in sh.decrypt_single(sh.parties_for_key(key), sh.ciphertexts(key, cts)[4])] == [msgs[4]] * 3
        # a party with a wrong share -> every receiver raises ValueError (PSK:119-123)
        parties = sh.parties_for_key(key)
        parties[1].secret_key.share.shares[2] += 1
        res = sh.decrypt_sequence(parties, sh.ciphertexts(key, cts[:4]), return_exceptions=True)
        assert all(isinstance(x, ValueError) for x in res)
        # a party that sends too few partial decryptions -> KeyError for the others (PSK:108-110)
        parties = sh.parties_for_key(key, wire=lambda s, v: v[:2] if s == "p2" else v)
        res = sh.decrypt_sequence(parties, sh.ciphertexts(key, cts[:4]), return_exceptions=True)
        assert isinstance(res[0], KeyError) and isinstance(res[2], KeyError) and [e.value for e in res[1]] == msgs[:4]
    finally:
        patch.uninstall()


def test_patched_compute_modulus_on_the_hip_engine(eng):
    """The stand-in's keygen loop, unpatched (CPython + sympy) and patched (HIP engine), from the same seed:
    the same biprime.  Candidates, generators and exchanged shares are identical in both runs, so every
    sieve verdict, Jacobi selection, v value and slot test of the patched run agreed with the CPU run."""
    from protocols.distributed_keygen_amd import patch

    for seed, key_length, batch in ((3, 128, 48), (4, 256, 64)):
        base = sh.keygen(seed=seed, key_length=key_length, batch_size=batch)
        assert len(set(base)) == 1
        f = sympy.factorint(base[0]) if key_length <= 128 else None
        assert f is None or (len(f) == 2 and all(e == 1 for e in f.values()))
        patch.install(engine=eng, package=sh.PACKAGE)
        try:
            got = sh.keygen(seed=seed, key_length=key_length, batch_size=batch)
        finally:
            patch.uninstall()
        assert got == base, (seed, key_length)


def test_rebound_leaf_runs_the_standins_own_scalar_methods_on_the_engine(eng):
    from protocols.distributed_keygen_amd import patch, synthetic

    psk, dk = sh.modules()
    key = synthetic.make_key(512, 3, 1)
    rng = random.Random(9)
    cts = [synthetic.encrypt(key, m, rng) for m in (5, 6)]
    orig = (psk.pow_mod, psk.mod_inv, dk.pow_mod)
    patch.install(engine=eng, package=sh.PACKAGE, scalars=False, leaf=True)
    try:
        assert psk.pow_mod is not orig[0] and dk.pow_mod is not orig[2]
        parties = sh.parties_for_key(key)              # the stand-in's constructor: mod_inv(theta, n) on the engine
        assert parties[0].secret_key.theta_inv == pow(key.theta, -1, key.n)
        got = sh.decrypt_sequence(parties, sh.ciphertexts(key, cts))
        assert [[e.value for e in r] for r in got] == [[5, 6]] * 3
        assert psk.pow_mod(7, -3, 10403) == pow(7, -3, 10403)
        # the stand-in's OWN scalar partial_decrypt (scalars=False left it in place): its pow_mod / mod_inv are the engine's
        e2 = key.exponent(2)
        want = pow(cts[0], e2, key.n_square) if e2 >= 0 else pow(pow(cts[0], -1, key.n_square), -e2, key.n_square)
        assert parties[1].secret_key.partial_decrypt(sh.ciphertexts(key, cts[:1])[0]) == want
    finally:
        patch.uninstall()
    assert (psk.pow_mod, psk.mod_inv, dk.pow_mod) == orig
