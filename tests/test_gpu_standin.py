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


def test_concurrent_single_decrypts_are_coalesced_on_the_hip_engine(eng):
    """3 co-located parties x 256 concurrent `decrypt()` coroutines at key_length 2048.  Unpatched
    shape: 768 lone launches of ~13 ms each.  Coalesced: per party ONE modexp launch (the three parties' launches side
    by side: same N, different exponents) and ONE recombination launch — bit-identical plaintexts, per-coroutine
    exceptions, well under 60 ms of wall time per burst once the per-key plans exist."""
    import time

    from protocols.distributed_keygen_amd import patch, synthetic

    key = synthetic.make_key(2048, 3, 1)
    rng = random.Random(2048)
    count = 256
    msgs = [0, 1, key.n - 1] + [rng.randrange(key.n) for _ in range(count - 3)]
    cts, seen = [], set()
    while len(cts) < count:                                  # distinct leading bits: the message id is made of them
        c = synthetic.encrypt(key, msgs[len(cts)], rng)
        tag = bin(c).zfill(32)[2:34]
        if tag not in seen:
            seen.add(tag)
            cts.append(c)
    patch.install(engine=eng, package=sh.PACKAGE)
    try:
        parties = sh.parties_for_key(key)
        got = sh.decrypt_many(parties, sh.ciphertexts(key, cts))        # first burst: prepares the per-key plans
        assert [e.value for e in got] == msgs * 3 and all(type(e.value) is int for e in got)
        stats = patch.coalescer(sh.PACKAGE).stats
        assert stats["partial_launches"] == 3 and stats["combine_launches"] == 3 and stats["largest_batch"] == count
        walls = []
        for _ in range(3):
            cobjs = sh.ciphertexts(key, cts)
            t0 = time.perf_counter()
            got = sh.decrypt_many(parties, cobjs)
            walls.append(time.perf_counter() - t0)
            assert [e.value for e in got] == msgs * 3 and all(not c.fresh for c in cobjs)
        stats = patch.coalescer(sh.PACKAGE).stats
        assert stats["partial_launches"] == 12 and stats["combine_launches"] == 12          # <= 3 launches per party and burst
        print(f"3 parties x {count} concurrent decrypt(): {[round(w * 1e3, 1) for w in walls]} ms per burst (inside the suite's process)")
        assert min(walls) < 0.25, walls            # 768 lone launches would be ~10 s; the strict bound is checked in a fresh process below
        # per-coroutine errors: one tampered partial decryption on the wire poisons exactly the recombinations that use it
        victim_tag = bin(cts[7]).zfill(32)[2:34]
        hub = parties[0].pool.hub
        orig_deliver = type(parties[1].pool)._deliver

        def tamper(self, to, message, msg_id):
            if self.me == "p2" and msg_id.endswith("#" + victim_tag):
                message = dict(message, value=(message["value"] + 1) % key.n_square)
            orig_deliver(self, to, message, msg_id)

        type(parties[1].pool)._deliver = tamper
        try:
            res = sh.decrypt_many(parties, sh.ciphertexts(key, cts[:16]), return_exceptions=True)
        finally:
            type(parties[1].pool)._deliver = orig_deliver
        for p in range(3):
            for k in range(16):
                r = res[p * 16 + k]
                if k == 7 and p != 1:             # parties 1 and 3 received p2's tampered share for ciphertext 7
                    assert isinstance(r, ValueError) and "not divisible by N" in str(r)
                else:
                    assert r.value == msgs[k]
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


def test_colocated_parties_share_the_launches_of_a_keygen_round(eng):
    """three parties in one process (the reference's distributed=False shape) generate a key_length
    1024 modulus with 1024 candidates per round.  With the per-process batcher every round's reconstruct + sieve,
    v-calculation and verdict run ONCE for all parties (one launch per kernel instead of three; the v-calculation with
    the parties' candidate groups concatenated); the time the engine is busy per round at least halves; the modulus is
    the same."""
    from protocols.distributed_keygen_amd import patch

    runs = {}
    for merge in (False, True):
        patch.install(engine=eng, package=sh.PACKAGE)
        try:
            rc = patch.round_coalescer(sh.PACKAGE)
            rc.merge = merge
            got = sh.keygen(seed=12, key_length=1024, batch_size=1024, prime_threshold=2000, correct_param=40)
            st = dict(rc.stats)
        finally:
            patch.uninstall()
        assert len(set(got)) == 1
        rounds = st["sieve_requests"] // 3
        runs.setdefault(merge, []).append((got[0], rounds, st))
    (n_a, rounds_a, st_a), (n_b, rounds_b, st_b) = min(runs[False], key=lambda r: r[2]["busy_s"]), min(runs[True], key=lambda r: r[2]["busy_s"])
    assert n_a == n_b and rounds_a == rounds_b
    assert st_a["sieve_launches"] == 3 * rounds_a and st_b["sieve_launches"] == rounds_b
    assert st_a["v_launches"] == 3 * st_b["v_launches"] and st_a["verdict_launches"] == 3 * st_b["verdict_launches"]
    per_round_a, per_round_b = st_a["busy_s"] / rounds_a, st_b["busy_s"] / rounds_b
    print(f"keygen K=1024, 1024 candidates/round, 3 co-located parties, {rounds_a} rounds: engine busy per round "
          f"{per_round_a * 1e3:.1f} ms one launch per party -> {per_round_b * 1e3:.1f} ms shared launches")
    # The factor (<= 0.5 x) is asserted in a fresh process below.  Here, late in a process that has used dozens of streams, the
    # two runs are one measurement each and came out within 2 % of each other either way round on different boxes: only a
    # shared round that is clearly SLOWER than separate launches fails.
    assert per_round_b < 1.25 * per_round_a, (per_round_a, per_round_b)


@pytest.mark.timeout(900)
def test_coalescing_wall_clock_in_a_fresh_process():
    """The two wall-clock claims of VERDICT r04 items 2 and 6, measured where a user would see them — a process of its own
    (tests/coalesce_timing_worker.py; the suite's process has used dozens of streams and is time-sliced): 3 parties x 256
    concurrent decrypt() in <= 60 ms per burst, and a co-located keygen round at <= 0.5 x the engine time."""
    import json
    import subprocess
    import sys
    from pathlib import Path

    worker = Path(__file__).resolve().parent / "coalesce_timing_worker.py"
    r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, f"rc={r.returncode}\n{r.stdout[-800:]}\n{r.stderr[-3000:]}"
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    bursts = next(d["decrypt_bursts_ms"] for d in lines if "decrypt_bursts_ms" in d)
    separate, shared = next(d["keygen_busy_ms_per_round"] for d in lines if "keygen_busy_ms_per_round" in d)
    print(f"fresh process: decrypt bursts {bursts} ms; keygen engine time per round {separate} -> {shared} ms")
    assert min(bursts) <= 60.0, bursts
    assert shared <= 0.5 * separate, (separate, shared)


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
