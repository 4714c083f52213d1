"""End-to-end keygen hot path on the GPU (BASELINE.json configs[0] shape, key_length 128, 3 parties):
rounds of candidate batches -> sieve -> Jacobi filter -> v-values of every party -> verdict, exactly
the compute steps of compute_modulus (distributed_keygen.py:1252-1360) with the message exchanges
replaced by local variables, until a biprime is found; then a threshold-decryption round trip with a
key built on that modulus.  Every step runs through the batched mirrors on the device; the result
is checked with independent host primality tests and against the oracle's verdict."""

from __future__ import annotations

import math
import random

import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


def test_three_party_keygen_hot_path_finds_a_biprime():
    from protocols.distributed_keygen_amd import Engine, biprime, synthetic
    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

    eng = Engine()
    rng = random.Random(128)
    n_parties, key_length, nbip, batch = 3, 128, 40, 768
    primes = oracle.small_prime_list(2000)
    found = None
    tested = sieved = 0
    from protocols.distributed_keygen_amd import shamir
    import sympy

    # the Shamir field of the round (DK:647-651) and textbook sharings of degree t (p_i, q_i) and 2t (zero)
    t = 1
    field = int(sympy.nextprime(2 ** (2 * (key_length // 2 + math.ceil(math.log2(n_parties))))))

    def share_out(secret, degree):
        coeffs = [secret] + [rng.randrange(field) for _ in range(degree)]
        return {j: sum(c * pow(j, e, field) for e, c in enumerate(coeffs)) % field for j in range(1, n_parties + 1)}

    for _round in range(40):
        shares = [synthetic.candidate_shares(rng, n_parties, key_length // 2) for _ in range(batch)]
        moduli = [sum(p) * sum(q) for p, q in shares]
        if _round == 0:
            # DK:1262-1284 on the device: every party's share of every candidate modulus (p * q + zero in
            # the Shamir field), then the reconstruction of all candidates and their sieve in one pass
            p_sh = [[share_out(pi, t) for pi in ps] for ps, _ in shares]          # [cand][owner] -> {holder: share}
            q_sh = [[share_out(qi, t) for qi in qs] for _, qs in shares]
            z_sh = [[share_out(0, 2 * t) for _ in range(n_parties)] for _ in shares]
            n_shares = {}
            for j in range(1, n_parties + 1):                                         # party j sums what it holds
                pj = [sum(o[j] for o in cand) % field for cand in p_sh]
                qj = [sum(o[j] for o in cand) % field for cand in q_sh]
                zj = [sum(o[j] for o in cand) % field for cand in z_sh]
                n_shares[j] = shamir.mul_add_shares_batch(pj, qj, zj, field, eng)
                assert n_shares[j][:3] == [oracle.shamir_mul_add(a, b, c, field) for a, b, c in zip(pj[:3], qj[:3], zj[:3])]
            assert shamir.reconstruct_batch(n_shares, field, 2 * t, eng) == moduli
            bad, surviving = shamir.reconstruct_and_sieve_batch(n_shares, field, 2 * t, primes, eng)
            assert bad == [oracle.small_prime_divisors_test(primes, m) for m in moduli]
            assert surviving == {k: m for k, (m, b) in enumerate(zip(moduli, bad)) if not b}
        has_div = biprime.small_prime_divisors_test_batch(primes, moduli, eng)          # DK:1288-1292
        surv = [k for k, bad in enumerate(has_div) if not bad]
        sieved += batch - len(surv)
        if not surv:
            continue
        mods = [moduli[k] for k in surv]
        # jointly random g: every party draws 4*40 values, summed mod N (DK:1028-1053)
        g_values = [[sum(rng.randint(0, m) for _ in range(n_parties)) % m for _ in range(4 * nbip)] for m in mods]
        v_all = [dict() for _ in surv]
        for i in range(1, n_parties + 1):                                                 # DK:1313-1329 per party
            vs = biprime.biprime_test_v_calculation_batch(
                g_values, i, mods, [shares[k][0][i - 1] for k in surv], [shares[k][1][i - 1] for k in surv], nbip, eng
            )
            for slot, v in zip(v_all, vs):
                slot[i] = v
        verdicts = biprime.biprime_test_with_v_i_batch(v_all, mods, nbip, eng, errors="return")   # DK:1339-1360
        tested += len(surv)
        for k, verdict, vd in zip(surv, verdicts, v_all):
            if isinstance(verdict, Exception):
                continue
            assert verdict == oracle.biprime_test_with_v_i(vd, moduli[k], nbip)
            if verdict:
                found = (shares[k], moduli[k])
                break
        if found:
            break
    assert found is not None, f"no biprime in {tested} tested / {sieved} sieved candidates"
    (p_parts, q_parts), n = found
    p, q = sum(p_parts), sum(q_parts)
    assert p * q == n and synthetic.is_probable_prime(p, rng) and synthetic.is_probable_prime(q, rng)
    assert sieved > 5 * tested                      # the sieve removes the bulk, as in the reference's counters

    # threshold decryption on the freshly generated modulus (structure of DK:1364-1500)
    n_fac = math.factorial(n_parties)
    lam, beta = n - p - q + 1, rng.randrange(n)
    bound = n_fac**2 * (1 << 40) * n * n_parties
    fl = [n_fac * lam, rng.randrange(-bound, bound)]
    fb = [n_fac * beta, rng.randrange(-bound, bound)]
    ev = lambda f, x: f[0] + f[1] * x  # noqa: E731
    share_vals = {i: ev(fl, i) * ev(fb, i) for i in (1, 2, 3)}
    theta = lam * beta * n_fac**3 % n
    if math.gcd(theta, n) != 1:
        pytest.skip("theta not invertible for this seed")
    keys = {i: GpuPaillierSharedKey(n, 1, i, ShareView({i: share_vals[i]}, 2, n_fac), theta, engine=eng) for i in (1, 2, 3)}
    msgs = [0, 1, n - 1, 31337, rng.randrange(n)]
    cts = eng.encrypt_batch(msgs, [rng.randrange(1, n) for _ in msgs], n)
    parts = {i: k.partial_decrypt_batch([PlainCiphertext(c, n) for c in cts]) for i, k in keys.items()}
    assert keys[2].decrypt_batch([{i: parts[i][e] for i in keys} for e in range(len(cts))]) == msgs


def test_biprime_round_with_state_on_the_device_equals_the_list_level_steps():
    """biprime.BiprimeRound (what patch.compute_modulus runs per round: the survivors' moduli and this party's v rows
    stay on the device between reconstruct + sieve, v-calculation and verdicts) against the list-level functions and the
    oracle, at key_length 1024 with 5 parties: same sieve verdicts, same v values, same verdicts with planted biprimes;
    a tampered copy of this party's values must NOT be answered from the device rows; a candidate with a short v list
    behaves as in the reference (False at its first failing slot, KeyError if it runs out of slots first)."""
    import sympy

    from protocols.distributed_keygen_amd import Engine, biprime, shamir, synthetic

    eng = Engine()
    rng = random.Random(2024)
    n_parties, t, key_length, nbip, batch = 5, 2, 1024, 40, 1500
    half = key_length // 2
    primes = oracle.small_prime_list(2000)
    field = int(sympy.nextprime(1 << (2 * (half + 4) + 44)))
    shares = [synthetic.candidate_shares(rng, n_parties, half) for _ in range(batch)]
    planted = synthetic.biprime_candidate_shares(rng, n_parties, half) if hasattr(synthetic, "biprime_candidate_shares") else None
    if planted is not None:
        shares[7] = planted
    moduli = [sum(p) * sum(q) for p, q in shares]
    points = list(range(1, n_parties + 1))
    by_party = {i: [] for i in points}
    for m in moduli:
        coeffs = [m] + [rng.randrange(field) for _ in range(2 * t)]
        for i in points:
            by_party[i].append(sum(c * pow(i, e, field) for e, c in enumerate(coeffs)) % field)
    rnd = biprime.BiprimeRound(eng)
    surviving = rnd.reconstruct_and_sieve(by_party, field, 2 * t, primes, points=points)
    bad, surviving_ref = shamir.reconstruct_and_sieve_batch(by_party, field, 2 * t, primes, eng, points=points)
    assert rnd.has_divisor == bad == [oracle.small_prime_divisors_test(primes, m) for m in moduli]
    assert surviving == surviving_ref == {k: moduli[k] for k in range(batch) if not bad[k]}
    surv = rnd.survivors
    assert len(surv) >= 5 and rnd.moduli == [moduli[k] for k in surv]
    mods = rnd.moduli
    g_values = [[rng.randrange(m) for _ in range(4 * nbip)] for m in mods]
    g_values[1] = g_values[1][:9]                                   # a candidate that runs out of generators
    ps = {i: [shares[k][0][i - 1] for k in surv] for i in points}
    qs = {i: [shares[k][1][i - 1] for k in surv] for i in points}
    v = {i: biprime.biprime_test_v_calculation_batch(g_values, i, mods, ps[i], qs[i], nbip, eng) for i in points}
    for index in (1, 3):
        own = rnd.v_calculation(g_values, index, ps[index], qs[index], nbip)
        assert own == v[index]
        e0 = biprime.biprime_exponent(index, mods[0], ps[index][0], qs[index][0])
        keep = [g for g in g_values[0] if oracle.jacobi_symbol(g, mods[0]) == 1][:nbip]
        assert own[0] == [pow(g, e0, mods[0]) for g in keep]
        v_by = [{i: list(v[i][c]) for i in points} for c in range(len(mods))]
        want = biprime.biprime_test_with_v_i_batch(v_by, mods, nbip, eng, errors="return")
        got = rnd.verdicts(v_by, nbip, errors="return")
        assert [type(x) for x in got] == [type(x) for x in want]
        assert [x for x in got if not isinstance(x, Exception)] == [x for x in want if not isinstance(x, Exception)]
        for c in (0, 1, 2, len(mods) - 1):            # candidate 1 ran out of generators: False at a failing slot, else KeyError
            try:
                assert got[c] == oracle.biprime_test_with_v_i(v_by[c], mods[c], nbip)
            except KeyError:
                assert isinstance(got[c], KeyError)
        # a tampered copy of this party's values: the verdicts follow the values handed in, not the device rows
        v_bad = [{i: list(vals) for i, vals in vc.items()} for vc in v_by]
        v_bad[0][index][0] = (v_bad[0][index][0] + 1) % mods[0]
        got_bad = rnd.verdicts(v_bad, nbip, errors="return")
        assert got_bad[0] == oracle.biprime_test_with_v_i(v_bad[0], mods[0], nbip)
        assert got_bad[2:] == got[2:]
