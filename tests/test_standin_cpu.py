"""CPU: the stand-in package (tests/standin) is a faithful enough target for patch.install() — patched with the
test double of the engine it returns what the unpatched stand-in computes on its own.  The GPU test
(test_gpu_standin.py) then runs the same flows on the HIP engine."""

from __future__ import annotations

import random

import pytest
import sympy

import standin_harness as sh
from fake_engine import FakeEngine


@pytest.fixture()
def key():
    from protocols.distributed_keygen_amd import synthetic

    return synthetic.make_key(128, 3, 1)


def test_standin_decrypt_sequence_patched_equals_unpatched(key):
    from protocols.distributed_keygen_amd import codec, patch, synthetic

    rng = random.Random(5)
    msgs = [0, 1, key.n - 1, 31337, 2**90 + 3]
    cts = [synthetic.encrypt(key, m, rng) for m in msgs]
    base = sh.decrypt_sequence(sh.parties_for_key(key), sh.ciphertexts(key, cts))
    assert [[e.value for e in r] for r in base] == [msgs] * 3
    eng = FakeEngine()
    patch.install(engine=eng, package=sh.PACKAGE)
    try:
        # party 2's list travels in wire form, the others' as plain ints
        wire = lambda sender, vals: [codec.encode_int(v) for v in vals] if sender == "p2" else vals
        got = sh.decrypt_sequence(sh.parties_for_key(key, wire=wire), sh.ciphertexts(key, cts))
        assert [[e.value for e in r] for r in got] == [msgs] * 3
        assert [e.value for e in sh.decrypt_single(sh.parties_for_key(key), sh.ciphertexts(key, cts)[3])] == [msgs[3]] * 3
        assert sorted(c for c in eng.calls if c[0] in ("powmod_batch", "combine_batch"))[:2] == [("combine_batch", 1), ("combine_batch", 1)]
    finally:
        patch.uninstall()
    assert [[e.value for e in r] for r in sh.decrypt_sequence(sh.parties_for_key(key), sh.ciphertexts(key, cts))] == [msgs] * 3


def test_standin_keygen_patched_equals_unpatched():
    from protocols.distributed_keygen_amd import patch

    base = sh.keygen(seed=21, key_length=64, batch_size=24)
    assert len(set(base)) == 1
    f = sympy.factorint(base[0])
    assert len(f) == 2 and all(e == 1 for e in f.values()) and all(p % 4 == 3 for p in f)
    eng = FakeEngine()
    patch.install(engine=eng, package=sh.PACKAGE)
    try:
        got = sh.keygen(seed=21, key_length=64, batch_size=24)
    finally:
        patch.uninstall()
    assert got == base
    assert {"shamir_reconstruct_sieve_batch", "biprime_verdict_batch"} <= {c[0] for c in eng.calls}
