"""Drop-in check against the REAL reference modules (build container only: needs /root/reference).

The unmodified reference ``paillier_shared_key.py`` / ``distributed_keygen.py`` are imported with
stand-ins for their un-vendored dependencies (the recipe of tests/golden/make_golden.py), three
in-process parties are wired through an in-memory pool, and the reference's own
``_decrypt_sequence_raw`` coroutine is run before and after ``patch.install()``: same plaintexts,
but one modexp launch and one recombination launch per party instead of one per ciphertext.
A test double of the engine is injected, so this runs without a GPU.
"""

from __future__ import annotations

import asyncio
import random
import sys
from pathlib import Path

import pytest

REF = Path("/root/reference/src/tno/mpc/protocols/distributed_keygen")
pytestmark = pytest.mark.skipif(not REF.exists(), reason="reference sources only exist in the build container")

sys.path.insert(0, str(Path(__file__).resolve().parent / "golden"))


class Hub:
    def __init__(self, names):
        self.names = names
        self.box = {n: {} for n in names}
        self.events = {n: {} for n in names}

    def pool(self, me):
        return FakePool(self, me)


class FakePool:
    """The three calls of tno.mpc.communication.Pool the decrypt path uses (DK:356-375, 476-497)."""

    def __init__(self, hub, me):
        self.hub, self.me = hub, me
        self.pool_handlers = {n: None for n in hub.names if n != me}

    def async_broadcast(self, message, msg_id=None, handler_names=None):
        for n in (handler_names if handler_names is not None else self.pool_handlers):
            self.hub.box[n].setdefault(msg_id, []).append((self.me, message))

    async def recv_all(self, msg_id=None):
        while len(self.hub.box[self.me].get(msg_id, [])) < len(self.pool_handlers):
            await asyncio.sleep(0)
        return tuple(self.hub.box[self.me].pop(msg_id))


@pytest.fixture(scope="module")
def ref():
    import make_golden

    psk, dk = make_golden.load_reference()
    return psk, dk, make_golden


def _parties(ref, engine_key):
    psk, dk, mg = ref
    shamir = sys.modules["tno.mpc.encryption_schemes.shamir"]
    key = mg.synth_key(random.Random(77), 128, 3, 1)
    names = ["p1", "p2", "p3"]
    hub = Hub(names)
    out = []
    for i, me in enumerate(names, start=1):
        share = shamir.IntegerShares(shamir._Scheme(3), {i: key["shares"][i]}, key["degree"], key["n_fac"] ** 2)
        dp = object.__new__(dk.DistributedPaillier)
        dp.secret_key = psk.PaillierSharedKey(n=key["n"], t=1, player_id=i, share=share, theta=key["theta"])
        dp.pool = hub.pool(me)
        dp.index = i
        dp.party_indices = {("self" if n == me else n): k for k, n in enumerate(names, start=1)}
        dp.session_id = 4242
        out.append(dp)
    return key, out


def _ciphertexts(ref, key, msgs):
    _, _, mg = ref
    pail = sys.modules["tno.mpc.encryption_schemes.paillier"]
    rng = random.Random(3)
    scheme = pail._PKScheme(key["n"])
    return [pail.PaillierCiphertext(mg.encrypt(rng, m, key["n"]), scheme) for m in msgs]


def test_decrypt_sequence_is_drop_in_and_batched(ref):
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import patch

    msgs = [0, 1, 7, 123456789, 2**100 + 5]
    key, parties = _parties(ref, None)
    cts = _ciphertexts(ref, key, msgs)

    async def run():
        return await asyncio.gather(*[dp._decrypt_sequence_raw(list(cts)) for dp in parties])

    base = asyncio.run(run())
    assert [[e.value for e in r] for r in base] == [msgs] * 3      # the reference alone

    eng = FakeEngine()
    patch.install(engine=eng)
    try:
        key, parties = _parties(ref, None)
        got = asyncio.run(run())
        assert [[e.value for e in r] for r in got] == [msgs] * 3
        assert all(type(e.value) is int for r in got for e in r)
        # per party: one modexp batch of 5 and one recombination batch of 5
        main_calls = [c for c in eng.calls if c[0] != "modinv_batch"]      # + one inversion batch for a negative exponent
        assert sorted(main_calls) == sorted([("powmod_batch", 5), ("combine_batch", 5)] * 3)
        # inversions: theta of each key once (PSK:50), and one batch of 5 for the party with a negative exponent
        assert all(c in (("modinv_batch", 5), ("modinv_batch", 1)) for c in eng.calls if c[0] == "modinv_batch")
        # single-ciphertext path (DK:314-382) still works through the patched scalar methods
        eng.calls.clear()

        async def one():
            return await asyncio.gather(*[dp._decrypt_raw(cts[3]) for dp in parties])

        assert [e.value for e in asyncio.run(one())] == [msgs[3]] * 3
        # receivers: only p1 ("self" for party 1) receives; others return None (DK:341-343, 516)
        async def recv():
            return await asyncio.gather(
                parties[0]._decrypt_sequence_raw(list(cts), ["self"]),
                parties[1]._decrypt_sequence_raw(list(cts), ["p1"]),
                parties[2]._decrypt_sequence_raw(list(cts), ["p1"]),
            )

        r = asyncio.run(recv())
        assert [e.value for e in r[0]] == msgs and r[1] is None and r[2] is None
        # a party holding a wrong share -> inconsistent partials -> ValueError like PSK:119-123
        # (same ciphertexts everywhere, so the message ids of DK:469-475 still match)
        parties[1].secret_key.share.shares[2] += 1
        parties[1].secret_key._mx_gpu_key = None

        async def bad():
            return await asyncio.gather(*[dp._decrypt_sequence_raw(list(cts)) for dp in parties], return_exceptions=True)

        assert all(isinstance(x, ValueError) for x in asyncio.run(bad()))
    finally:
        patch.uninstall()
    key, parties = _parties(ref, None)
    assert [[e.value for e in r] for r in asyncio.run(run())] == [msgs] * 3   # originals restored


def test_patched_keygen_classmethods_match_originals(ref, golden_biprime):
    from conftest import unhex
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import patch

    _, dk, _ = ref
    DP = dk.DistributedPaillier
    names = ["small_prime_divisors_test", "biprime_test_v_calculation", "biprime_test_with_v_i"]
    orig = {n: getattr(DP, "_DistributedPaillier__" + n) for n in names}
    cand = next(c for c in golden_biprime["candidates"] if c["label"] == "k128_n3_biprime0")
    comp = next(c for c in golden_biprime["candidates"] if c["label"] == "k128_n3_composite2")
    patch.install(engine=FakeEngine())
    try:
        new = {n: getattr(DP, "_DistributedPaillier__" + n) for n in names}
        primes = [3, 5, 7, 11, 13]
        for m in (15, 77, 221, 10403):
            assert new["small_prime_divisors_test"](primes, m) == orig["small_prime_divisors_test"](primes, m)
        for c in (cand, comp):
            modulus = unhex(c["modulus"])
            gs = [unhex(g) for g in c["g_values"]]
            pi = {f"party{i}": i for i in (1, 2, 3)}
            b_new = dk.Batched(dk.AdditiveVariable(label="v", modulus=modulus), batch_size=40)
            b_old = dk.Batched(dk.AdditiveVariable(label="v", modulus=modulus), batch_size=40)
            for i in (1, 2, 3):
                p_i, q_i = unhex(c["p_parts"][i - 1]), unhex(c["q_parts"][i - 1])
                a = new["biprime_test_v_calculation"](gs, i, modulus, p_i, q_i, 40)
                b = orig["biprime_test_v_calculation"](gs, i, modulus, p_i, q_i, 40)
                va = [v._sharing.get(i) for v in a.variables]
                assert va == [v._sharing.get(i) for v in b.variables]
                vals = [x for x in va if x is not None]
                b_new.set_share(i, vals)
                b_old.set_share(i, vals)
            assert new["biprime_test_with_v_i"](b_new, modulus, 40, pi) == orig["biprime_test_with_v_i"](b_old, modulus, 40, pi) == c["verdict"]
    finally:
        patch.uninstall()
    assert getattr(DP, "_DistributedPaillier__small_prime_divisors_test")([3], 9) is True


class FullPool(FakePool):
    """+ the point-to-point call the keygen path uses (utils.py:528-553)."""

    def asend(self, party, message, msg_id=None):
        self.hub.box[party].setdefault(msg_id, []).append((self.me, message))


def _run_compute_modulus(ref, seed, batch_size):
    """Three in-process parties run the reference class-method DistributedPaillier.compute_modulus
    (patched or not) over an in-memory pool with seeded randomness."""
    import random as _random
    import secrets as _secrets

    _, dk, _ = ref
    DP = dk.DistributedPaillier
    rng = _random.Random(seed)
    saved = (_secrets.randbits, _secrets.randbelow, dk.randint)
    _secrets.randbits = rng.getrandbits
    _secrets.randbelow = lambda n: rng.randrange(n)
    dk.secrets.randbits = rng.getrandbits
    dk.randint = rng.randint
    try:
        names = ["p1", "p2", "p3"]
        hub = Hub(names)
        key_length, t = 64, 1

        async def party(i, me):
            pool = FullPool(hub, me)
            party_indices = {("self" if n == me else n): k for k, n in enumerate(names, start=1)}
            n_players, prime_length, prime_list, sh_t, sh_2t, shares = DP.setup_input(pool, key_length, 200, t)
            return await DP.compute_modulus(
                shares, i, pool, prime_list, party_indices, prime_length, sh_t, sh_2t, 20, 7, batch_size
            )

        async def run():
            return await asyncio.gather(*[party(i, me) for i, me in enumerate(names, start=1)])

        return asyncio.run(run())
    finally:
        _secrets.randbits, _secrets.randbelow, dk.randint = saved
        dk.secrets.randbits = saved[0]


def test_compute_modulus_is_drop_in(ref):
    """The reference's own keygen loop (compute_modulus, DK:1211-1362) with and without the patch:
    same seeded candidates -> the same biprime modulus, found with one launch per step and round."""
    import sympy

    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import patch

    base = _run_compute_modulus(ref, seed=11, batch_size=40)
    assert len(set(base)) == 1
    n = base[0]
    f = sympy.factorint(n)
    assert len(f) == 2 and all(e == 1 for e in f.values())            # a true biprime
    eng = FakeEngine()
    patch.install(engine=eng)
    try:
        got = _run_compute_modulus(ref, seed=11, batch_size=40)
        stats = dict(patch.round_coalescer().stats)
    finally:
        patch.uninstall()
    assert got == base
    # the three parties share this process (the reference's distributed=False shape): every round's reconstruct + sieve,
    # v-calculation (the parties' candidate groups concatenated: same moduli and generators, own exponents) and verdicts
    # run ONCE for all of them (coalesce.RoundCoalescer) — VERDICT r04 item 6
    assert stats["sieve_requests"] == 3 * stats["sieve_launches"] and stats["sieve_launches"] >= 1
    assert stats["v_requests"] == 3 * stats["v_launches"] and stats["verdict_requests"] == 3 * stats["verdict_launches"]
    survivors_per_round = [c[1] // 3 for c in eng.calls if c[0] == "biprime_v_batch"]
    assert all(c[1] % 3 == 0 for c in eng.calls if c[0] == "biprime_v_batch") and len(survivors_per_round) == stats["v_launches"]
    kinds = {c[0] for c in eng.calls}
    # one round = biprime.BiprimeRound: reconstruct + sieve, v-calculation and verdicts as one call each per party, the
    # survivors' moduli and the party's own v rows handed from step to step (the double checks that the kept moduli rows
    # belong to the candidates of the later calls) — and the exchanged own column, being what was computed, is taken
    # "from the device"
    assert {"shamir_reconstruct_sieve_batch", "biprime_v_batch", "biprime_verdict_columns", "biprime_verdict_batch",
            "own_column_from_device"} <= kinds
    rounds = sum(1 for c in eng.calls if c[0] == "shamir_reconstruct_sieve_batch")
    # N reconstruction (DK:1284) + sieve (DK:1288-1292) of a whole round in one call
    assert all(c[1] == 40 for c in eng.calls if c[0] == "shamir_reconstruct_sieve_batch")
    assert sum(1 for c in eng.calls if c[0] == "biprime_v_batch") <= rounds
    assert sum(1 for c in eng.calls if c[0] == "own_column_from_device") == sum(1 for c in eng.calls if c[0] == "biprime_verdict_columns")


def test_single_decrypt_interleaved_with_pending_sequence(ref):
    """a single decrypt() running concurrently with a decrypt_sequence() on the
    same scheme.  Party 1 starts a 3-element sequence and a single decryption; the other parties
    answer the SINGLE first, so party 1's `_decrypt_raw` reaches PaillierSharedKey.decrypt while
    its `_decrypt_sequence_raw` is still waiting in recv_all.  Both must return plaintext ints."""
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import patch

    msgs = [11, 22, 33]
    single = 44
    patch.install(engine=FakeEngine())
    try:
        key, parties = _parties(ref, None)
        cts = _ciphertexts(ref, key, msgs + [single])
        seq_cts, single_ct = cts[:3], cts[3]

        async def party1():
            seq_task = asyncio.ensure_future(parties[0]._decrypt_sequence_raw(list(seq_cts)))
            one_task = asyncio.ensure_future(parties[0]._decrypt_raw(single_ct))
            return await asyncio.gather(seq_task, one_task)

        async def other(dp):
            one = await dp._decrypt_raw(single_ct)            # answers the single first ...
            for _ in range(5):
                await asyncio.sleep(0)
            seq = await dp._decrypt_sequence_raw(list(seq_cts))   # ... and the sequence afterwards
            return seq, one

        async def run():
            return await asyncio.gather(party1(), other(parties[1]), other(parties[2]))

        for seq, one in asyncio.run(run()):
            assert type(one.value) is int and one.value == single
            assert [e.value for e in seq] == msgs and all(type(e.value) is int for e in seq)

        # two overlapping sequences and a single, all on party 1, every interleaving of the answers
        async def party1_three():
            a = asyncio.ensure_future(parties[0]._decrypt_sequence_raw(list(seq_cts)))
            b = asyncio.ensure_future(parties[0]._decrypt_sequence_raw([cts[1], cts[0]]))
            c = asyncio.ensure_future(parties[0]._decrypt_raw(single_ct))
            return await asyncio.gather(a, b, c)

        async def other_three(dp, order):
            # the three protocols run concurrently on every party; they are STARTED in different orders
            tasks = {}
            for what in order:
                if what == "a":
                    tasks["a"] = asyncio.ensure_future(dp._decrypt_sequence_raw(list(seq_cts)))
                elif what == "b":
                    tasks["b"] = asyncio.ensure_future(dp._decrypt_sequence_raw([cts[1], cts[0]]))
                else:
                    tasks["c"] = asyncio.ensure_future(dp._decrypt_raw(single_ct))
                for _ in range(3):
                    await asyncio.sleep(0)
            return await asyncio.gather(tasks["a"], tasks["b"], tasks["c"])

        async def run3():
            return await asyncio.gather(party1_three(), other_three(parties[1], "cba"), other_three(parties[2], "bca"))

        for a, b, c in asyncio.run(run3()):
            assert [e.value for e in a] == msgs and [e.value for e in b] == [22, 11] and c.value == single
    finally:
        patch.uninstall()


def test_patched_sequence_keeps_reference_error_behaviour(ref):
    """A party that sends too few partial decryptions leaves later ciphertexts without its share: the
    reference's loop raises KeyError at PSK:110 for them; so does the batched path."""
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import patch

    patch.install(engine=FakeEngine())
    try:
        key, parties = _parties(ref, None)
        cts = _ciphertexts(ref, key, [5, 6, 7])
        orig_broadcast = parties[1].pool.async_broadcast

        def short_broadcast(message, msg_id=None, handler_names=None):
            message = dict(message, value=message["value"][:2])
            orig_broadcast(message, msg_id=msg_id, handler_names=handler_names)

        parties[1].pool.async_broadcast = short_broadcast

        async def run():
            return await asyncio.gather(*[dp._decrypt_sequence_raw(list(cts)) for dp in parties], return_exceptions=True)

        res = asyncio.run(run())
        assert isinstance(res[0], KeyError) and isinstance(res[2], KeyError)      # parties 1 and 3 miss share 2
        assert [e.value for e in res[1]] == [5, 6, 7]                                # party 2 has everything
    finally:
        patch.uninstall()


def test_limits_are_checked_before_any_round(ref):
    from protocols.distributed_keygen_amd import patch

    patch.check_limits(None, [3, 5, 1999], 1024, 3)
    patch.check_limits(None, [3, 5, (1 << 21) + 7], 1024, 3)
    with pytest.raises(ValueError, match="prime_threshold"):
        patch.check_limits(None, [3, 5, (1 << 31) + 11], 1024, 3)
    patch.check_limits(None, [3, 5], 4096, 5)                    # key_length 8192 is inside the engine's range
    with pytest.raises(ValueError, match="key_length"):
        patch.check_limits(None, [3, 5], 4200, 3)


def test_install_can_leave_the_scalar_methods_to_the_reference(ref):
    """install(scalars=False): single-ciphertext decrypt() keeps the reference's own scalar path, the
    sequence path is batched."""
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import patch

    psk, dk, _ = ref
    orig_pd, orig_d = psk.PaillierSharedKey.partial_decrypt, psk.PaillierSharedKey.decrypt
    eng = FakeEngine()
    patch.install(engine=eng, scalars=False)
    try:
        assert psk.PaillierSharedKey.partial_decrypt is orig_pd and psk.PaillierSharedKey.decrypt is orig_d
        key, parties = _parties(ref, None)
        cts = _ciphertexts(ref, key, [9, 8, 7])

        async def one():
            return await asyncio.gather(*[dp._decrypt_raw(cts[0]) for dp in parties])

        eng.calls.clear()
        assert [e.value for e in asyncio.run(one())] == [9, 9, 9]
        assert not [c for c in eng.calls if c[0] in ("powmod_batch", "combine_batch")]      # the reference's own arithmetic

        async def seq():
            return await asyncio.gather(*[dp._decrypt_sequence_raw(list(cts)) for dp in parties])

        assert [[e.value for e in r] for r in asyncio.run(seq())] == [[9, 8, 7]] * 3
        assert sum(1 for c in eng.calls if c == ("powmod_batch", 3)) == 3
    finally:
        patch.uninstall()
    assert psk.PaillierSharedKey.partial_decrypt is orig_pd


def test_leaf_operators_can_be_rebound(ref):
    """install(leaf=True) rebinds pow_mod / mod_inv where the reference imported them by name
    (distributed_keygen.py:35, paillier_shared_key.py:20): the reference's own PaillierSharedKey
    constructor (mod_inv at :50) and scalar partial_decrypt (pow_mod at :92) then run on the engine."""
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import operators, patch

    psk, dk, mg = ref
    orig = (psk.pow_mod, psk.mod_inv, dk.pow_mod)
    eng = FakeEngine()
    patch.install(engine=eng, scalars=False, leaf=True)
    try:
        assert psk.pow_mod is not orig[0] and psk.mod_inv is not orig[1] and dk.pow_mod is not orig[2]
        eng.calls.clear()
        key, parties = _parties(ref, None)                       # the reference's constructor: mod_inv(theta, n)
        assert ("modinv_batch", 1) in eng.calls
        cts = _ciphertexts(ref, key, [41])
        eng.calls.clear()
        p = parties[0].secret_key.partial_decrypt(cts[0])        # the reference's own scalar method, GPU leaf
        assert ("powmod_batch", 1) in eng.calls
        from oracle import oracle

        assert p == oracle.partial_decrypt(cts[0].peek_value(), key["n"], 1, key["degree"], key["n_fac"], key["shares"][1])
        # moduli outside the engine's domain (even, < 3) fall through to the reference's own leaf: the rebound name is total
        assert psk.pow_mod(7, 5, 1 << 40) == pow(7, 5, 1 << 40) and psk.pow_mod(3, 4, 2) == 1 and psk.mod_inv(3, 16) == 11
        assert operators.pow_mod(7, -3, 101, engine=eng) == pow(7, -3, 101)
        assert operators.mod_inv(7, 101, engine=eng) == pow(7, -1, 101)
        assert operators.pow_mod_batch_multi([[2, 3], [5]], [10, 3], [101, 103], engine=eng) == [[pow(2, 10, 101), pow(3, 10, 101)], [pow(5, 3, 103)]]
    finally:
        patch.uninstall()
    assert (psk.pow_mod, psk.mod_inv, dk.pow_mod) == orig


def test_concurrent_single_decrypts_share_launches(ref):
    """The reference's own API shape `asyncio.gather(*(scheme.decrypt(c) for c in cs))` (its test
    test/test_distributed_keygen.py:132-158): every coroutine reaches PaillierSharedKey.partial_decrypt at DK:345-349 and
    .decrypt at DK:378-380 on its own.  Patched, the coroutines pending in the same turn of the event loop share ONE
    modexp batch and ONE recombination batch per party (coalesce.Coalescer) — same plaintexts, same per-coroutine
    exceptions, `get_value()` still called once per ciphertext."""
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import patch

    msgs = [(k * 7919 + 3) % 100003 for k in range(48)]
    key, parties = _parties(ref, None)
    # distinct leading bits per ciphertext: the reference's message id is made of them (DK:352-355)
    cts = _ciphertexts(ref, key, msgs)
    assert len({bin(c.peek_value()).zfill(32)[2:34] for c in cts}) == len(cts)

    async def run(ps, return_exceptions=False):
        return await asyncio.gather(*[dp._decrypt_raw(c) for dp in ps for c in cts], return_exceptions=return_exceptions)

    base = asyncio.run(run(parties))
    assert [e.value for e in base] == msgs * 3                         # the reference alone

    eng = FakeEngine()
    patch.install(engine=eng)
    try:
        key, parties = _parties(ref, None)
        cts = _ciphertexts(ref, key, msgs)
        assert all(c.fresh for c in cts) if hasattr(cts[0], "fresh") else True
        eng.calls.clear()
        got = asyncio.run(run(parties))
        assert [e.value for e in got] == msgs * 3 and all(type(e.value) is int for e in got)
        stats = patch.coalescer().stats
        # per party ONE modexp batch of all 48 and ONE recombination batch of all 48 — not 48 of each
        assert sorted(c for c in eng.calls if c[0] in ("powmod_batch", "combine_batch")) == sorted(
            [("powmod_batch", 48), ("combine_batch", 48)] * 3)
        assert stats["partial_launches"] == 3 and stats["combine_launches"] == 3 and stats["largest_batch"] == 48
        # --- per-coroutine exceptions
        pail = sys.modules["tno.mpc.encryption_schemes.paillier"]
        foreign = pail.PaillierCiphertext(cts[0].peek_value(), pail._PKScheme(key["n"] + 2))    # another key: PSK:67-68

        async def mixed():
            return await asyncio.gather(
                parties[0]._decrypt_raw(cts[1]), parties[0]._decrypt_raw(foreign), parties[0]._decrypt_raw("not a ciphertext"),
                parties[1]._decrypt_raw(cts[1]), parties[2]._decrypt_raw(cts[1]), return_exceptions=True)

        r = asyncio.run(mixed())
        assert r[0].value == msgs[1] and isinstance(r[1], ValueError) and isinstance(r[2], TypeError)
        assert r[3].value == r[4].value == msgs[1]
        # a party with a wrong share: every recombination that uses its partial raises ValueError (PSK:119-123) in the
        # coroutine that owns it, and nothing else is disturbed; with t = 1 all three partials are needed by everyone
        parties[1].secret_key.share.shares[2] += 1
        parties[1].secret_key._mx_gpu_key = None
        bad = asyncio.run(run(parties, return_exceptions=True))
        assert all(isinstance(x, ValueError) and "not divisible by N" in str(x) for x in bad)
        # a missing share raises KeyError (PSK:108-110) in its coroutine only
        parties[1].secret_key.share.shares[2] -= 1
        parties[1].secret_key._mx_gpu_key = None
        orig_recv = parties[0].pool.recv_all
        victim = f"distributed_decryption_session#{parties[0].session_id}_hash#{bin(cts[5].peek_value()).zfill(32)[2:34]}"

        async def recv_all(msg_id=None):
            out = await orig_recv(msg_id=msg_id)
            return tuple(m for m in out if not (msg_id == victim and m[0] == "p2"))

        parties[0].pool.recv_all = recv_all
        res = asyncio.run(run(parties, return_exceptions=True))
        assert isinstance(res[5], KeyError) and [e.value for k, e in enumerate(res) if k != 5] == (msgs * 3)[:5] + (msgs * 3)[6:]
    finally:
        patch.uninstall()
    assert patch.coalescer() is None
    key, parties = _parties(ref, None)
    cts = _ciphertexts(ref, key, msgs[:4])

    async def few():
        return await asyncio.gather(*[dp._decrypt_raw(c) for dp in parties for c in cts])

    assert [e.value for e in asyncio.run(few())] == msgs[:4] * 3       # originals restored


def test_coalescer_batches_only_what_is_pending_together():
    """Bursts that do not overlap get a launch each; a cancelled coroutine does not poison its batch; an engine
    failure reaches every coroutine of the batch (and only that batch)."""
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import synthetic
    from protocols.distributed_keygen_amd.coalesce import Coalescer
    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

    key = synthetic.make_key(64, 3, 1, kappa=20)
    eng = FakeEngine()
    gk = GpuPaillierSharedKey(key.n, 1, 1, ShareView({1: key.shares[1]}, key.degree, key.n_fac), key.theta, engine=eng)
    co = Coalescer(eng)
    rng = random.Random(1)
    cts = [PlainCiphertext(synthetic.encrypt(key, m, rng), key.n) for m in range(10)]
    want = [gk.partial_decrypt(PlainCiphertext(c.value, key.n)) for c in cts]

    async def scenario():
        first = await asyncio.gather(*[co.partial_decrypt(gk, c) for c in cts[:6]])
        assert co.stats["partial_launches"] == 1
        tasks = [asyncio.ensure_future(co.partial_decrypt(gk, c)) for c in cts[6:]]
        await asyncio.sleep(0)
        tasks[1].cancel()
        rest = await asyncio.gather(*tasks, return_exceptions=True)
        return first, rest

    eng.calls.clear()
    first, rest = asyncio.run(scenario())
    assert first == want[:6] and rest[0] == want[6] and isinstance(rest[1], asyncio.CancelledError) and rest[2:] == want[8:]
    assert [c for c in eng.calls if c[0] == "powmod_batch"] == [("powmod_batch", 6), ("powmod_batch", 4)]
    assert all(not c.fresh for c in cts)                                # get_value() once per ciphertext (PSK:69)

    def boom(*a, **k):
        raise RuntimeError("engine down")

    eng.powmod_nsquare_batch = boom

    async def failing():
        return await asyncio.gather(*[co.partial_decrypt(gk, c) for c in cts[:3]], return_exceptions=True)

    assert all(isinstance(x, RuntimeError) for x in asyncio.run(failing()))
    # linger: a time-based flush for pools whose messages arrive over several turns of the loop
    del eng.powmod_nsquare_batch
    slow = Coalescer(eng, linger=0.02)

    async def trickle():
        async def late(c, delay):
            await asyncio.sleep(delay)
            return await slow.partial_decrypt(gk, c)

        return await asyncio.gather(*[late(c, 0.002 * k) for k, c in enumerate(cts[:5])])

    assert asyncio.run(trickle()) == want[:5] and slow.stats["partial_launches"] == 1


def test_coalesced_failures_stay_with_the_coroutine_or_key_that_caused_them():
    """ADVICE r05 (medium).  PSK:89-91: under a negative Lagrange exponent the ciphertext is inverted modulo N^2 first, and
    in the reference only the decrypt() that holds a ciphertext without an inverse fails.  One device inversion per batch
    fails as a whole, so the coalescer must find the offender and give ValueError to ITS coroutine alone — not to the other
    ciphertexts of that key, not to the other keys of the burst; and a key whose batch the engine refuses must not take
    the co-located keys' batches with it."""
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import synthetic
    from protocols.distributed_keygen_amd.coalesce import Coalescer
    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

    key = synthetic.make_key(64, 3, 1, kappa=20)
    neg = next(i for i in (1, 2, 3) if key.exponent(i) < 0)
    pos = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
    eng = FakeEngine()
    mk = lambda i: GpuPaillierSharedKey(key.n, 1, i, ShareView({i: key.shares[i]}, key.degree, key.n_fac), key.theta, engine=eng)
    k_neg, k_pos = mk(neg), mk(pos)
    rng = random.Random(5)
    good = [synthetic.encrypt(key, m, rng) for m in (7, 8, 9)]
    ct = lambda v: PlainCiphertext(v, key.n)
    co = Coalescer(eng)

    async def burst():
        return await asyncio.gather(co.partial_decrypt(k_neg, ct(good[0])), co.partial_decrypt(k_neg, ct(0)),
                                    co.partial_decrypt(k_neg, ct(key.p * 5)), co.partial_decrypt(k_neg, ct(good[1])),
                                    co.partial_decrypt(k_pos, ct(good[2])), co.partial_decrypt(k_pos, ct(0)), return_exceptions=True)

    out = asyncio.run(burst())
    n2 = key.n_square
    assert out[0] == pow(pow(good[0], -1, n2), -key.exponent(neg), n2) == k_neg.partial_decrypt(ct(good[0]))
    assert isinstance(out[1], ValueError) and isinstance(out[2], ValueError) and out[1] is not out[2]
    assert out[3] == k_neg.partial_decrypt(ct(good[1]))
    assert out[4] == k_pos.partial_decrypt(ct(good[2])) and out[5] == 0          # no inversion under a positive exponent: 0^e = 0
    with pytest.raises(ValueError):
        k_neg.partial_decrypt(ct(0))                                             # the un-coalesced call fails the same way
    # every ciphertext of the key without an inverse: no modexp launch for that key, the other key unaffected
    async def all_bad():
        return await asyncio.gather(co.partial_decrypt(k_neg, ct(0)), co.partial_decrypt(k_pos, ct(good[0])), return_exceptions=True)

    bad_out = asyncio.run(all_bad())
    assert isinstance(bad_out[0], ValueError) and bad_out[1] == k_pos.partial_decrypt(ct(good[0]))

    # an engine that runs keys side by side and refuses ONE key's batch: only that key's coroutines see the failure
    class SideBySide(FakeEngine):
        def powmod_nsquare_groups(self, jobs):
            if any(n == other.n for _, _, n in jobs):
                raise ValueError("rows narrower than N^2")
            return [self.powmod_nsquare_batch(v, e, n) for v, e, n in jobs]

        def powmod_nsquare_batch(self, bases, exp, n, keep_rows=False):
            if n == other.n:
                raise ValueError("rows narrower than N^2")
            return super().powmod_nsquare_batch(bases, exp, n, keep_rows)

    other = synthetic.make_key(64, 3, 1, kappa=20, seed=99)
    eng2 = SideBySide()
    a = GpuPaillierSharedKey(key.n, 1, pos, ShareView({pos: key.shares[pos]}, key.degree, key.n_fac), key.theta, engine=eng2)
    opos = next(i for i in (1, 2, 3) if other.exponent(i) > 0)
    b = GpuPaillierSharedKey(other.n, 1, opos, ShareView({opos: other.shares[opos]}, other.degree, other.n_fac), other.theta, engine=eng2)
    co2 = Coalescer(eng2)

    async def two_keys():
        return await asyncio.gather(co2.partial_decrypt(a, ct(good[0])), co2.partial_decrypt(b, PlainCiphertext(12345, other.n)),
                                    co2.partial_decrypt(a, ct(good[1])), return_exceptions=True)

    r = asyncio.run(two_keys())
    assert r[0] == pow(good[0], key.exponent(pos), n2) and r[2] == pow(good[1], key.exponent(pos), n2)
    assert isinstance(r[1], ValueError) and "narrower" in str(r[1])
