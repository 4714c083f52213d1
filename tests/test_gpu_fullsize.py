"""GPU tests at BASELINE.json's full sizes through size-independent properties (the oracle is too
slow to recompute 10 000 4096-bit modexps): encrypt -> partial-decrypt -> recombine round trips,
additive homomorphism, biprimality of true biprimes vs composites — with samples cross-checked
bit-exactly against the oracle."""

from __future__ import annotations

import math
import random

import pytest

from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from protocols.distributed_keygen_amd import Engine

    return Engine()


def _decrypt_all(eng, key, cts):
    n2 = key.n_square
    partials = []
    for i in range(1, key.degree + 2):
        e = key.exponent(i)
        bases = cts if e >= 0 else eng.modinv_batch(cts, n2)
        partials.append(eng.powmod_nsquare_batch(bases, abs(e), key.n))     # the product's path (PSK:92)
    rows = [[partials[i][k] for i in range(key.degree + 1)] for k in range(len(cts))]
    msgs, ok = eng.combine_batch(rows, key.n, key.theta_inv)
    return partials, msgs, ok


def test_c3_10k_ciphertexts_roundtrip_key2048(eng):
    """configs[2]: 3-party, key_length 2048, t=1, 10 000 ciphertexts."""
    from protocols.distributed_keygen_amd import synthetic

    key = synthetic.make_key(2048, 3, 1)
    n, n2 = key.n, key.n_square
    rng = random.Random(2048)
    batch = 10000
    msgs = [rng.randrange(n) for _ in range(batch)]
    msgs[:4] = [0, 1, n - 1, n // 2]
    rs = [rng.randrange(1, n) for _ in range(batch)]
    rn = eng.powmod_batch(rs, n, n2)                                   # r^N mod N^2 on the device
    for k in (0, 1, batch // 2, batch - 1):
        assert rn[k] == oracle.pow_mod(rs[k], n, n2)
    cts = [(1 + m * n) % n2 * x % n2 for m, x in zip(msgs, rn)]
    partials, got, ok = _decrypt_all(eng, key, cts)
    assert all(ok) and got == msgs                                     # decrypt(encrypt(m)) == m, all 10 000
    # one party's partial decryptions, every 4th ciphertext, bit-exact vs pow() on all host cores
    import multiprocessing as mp

    i_pos = next(i for i in (1, 2, 3) if key.exponent(i) >= 0)
    idx = list(range(0, batch, 4))
    with mp.Pool() as pool:
        want = pool.starmap(pow, [(cts[k], key.exponent(i_pos), n2) for k in idx], chunksize=8)
    assert [partials[i_pos - 1][k] for k in idx] == want
    # the N-adic pair kernel (mx_powmod_nsquare, what partial_decrypt_batch and bench.py launch), in
    # both lane geometries, on all 10 000: identical to the generic-modulus kernel checked above
    for lpl in (9, 18):
        eng.set_limbs_per_lane(lpl)
        assert eng.powmod_nsquare_batch(cts, key.exponent(i_pos), n) == partials[i_pos - 1], lpl
    eng.set_limbs_per_lane(0)
    for k in (3, 4999, 9999):                                          # samples bit-exact vs the oracle
        for i in (1, 2, 3):
            assert partials[i - 1][k] == oracle.partial_decrypt(cts[k], n, i, key.degree, key.n_fac, key.shares[i])
    # additive homomorphism on the whole batch: Dec(c_k * c_{k+1}) = m_k + m_{k+1} mod N
    prod = [cts[k] * cts[(k + 1) % batch] % n2 for k in range(0, batch, 10)]
    _, got2, ok2 = _decrypt_all(eng, key, prod)
    assert all(ok2) and got2 == [(msgs[k] + msgs[(k + 1) % batch]) % n for k in range(0, batch, 10)]


def test_c5_key4096_roundtrip(eng):
    """configs[4] shape: key_length 4096 (8200-bit modulus, ~8300-bit exponent), 256 ciphertexts."""
    from protocols.distributed_keygen_amd import synthetic

    key = synthetic.make_key(4096, 3, 1)
    n, n2 = key.n, key.n_square
    assert eng.geometry(n2.bit_length())[0] == 32
    rng = random.Random(4096)
    msgs = [rng.randrange(n) for _ in range(256)]
    rs = [rng.randrange(1, n) for _ in range(256)]
    rn = eng.powmod_batch(rs, n, n2)
    assert rn[7] == oracle.pow_mod(rs[7], n, n2)
    cts = [(1 + m * n) % n2 * x % n2 for m, x in zip(msgs, rn)]
    partials, got, ok = _decrypt_all(eng, key, cts)
    assert all(ok) and got == msgs
    assert partials[1][5] == oracle.partial_decrypt(cts[5], n, 2, key.degree, key.n_fac, key.shares[2])


@pytest.mark.parametrize("key_length,n_parties,n_cands", [(1024, 3, 256), (2048, 5, 64)])
def test_c2_c4_biprimality_batches(eng, key_length, n_parties, n_cands):
    """configs[1]/[3] shape: candidate moduli x 40 Jacobi-1 bases; true biprimes pass every slot,
    composites fail; sieve agrees with the oracle; samples bit-exact."""
    from protocols.distributed_keygen_amd import biprime, synthetic

    rng = random.Random(key_length + n_parties)
    half = key_length // 2
    cands = []
    for c in range(n_cands):
        p_parts, q_parts = synthetic.candidate_shares(rng, n_parties, half)
        want_biprime = c % 32 == 0
        if want_biprime:                                   # steer the last shares so p and q are prime
            for parts in (p_parts, q_parts):
                base = sum(parts)
                t = base
                while not synthetic.is_probable_prime(t, rng, 8):
                    t += 4
                parts[-1] += t - base
        cands.append((p_parts, q_parts, sum(p_parts) * sum(q_parts), want_biprime))
    moduli = [c[2] for c in cands]
    primes = oracle.small_prime_list(2000)
    sieve = biprime.small_prime_divisors_test_batch(primes, moduli, eng)
    for k in range(0, n_cands, 17):
        assert sieve[k] == oracle.small_prime_divisors_test(primes, moduli[k])
    assert not any(s for s, c in zip(sieve, cands) if c[3])            # biprimes of large primes survive
    g_values = [[rng.randrange(m) for _ in range(160)] for m in moduli]
    v_all = [dict() for _ in cands]
    for i in range(1, n_parties + 1):
        got = biprime.biprime_test_v_calculation_batch(
            g_values, i, moduli, [c[0][i - 1] for c in cands], [c[1][i - 1] for c in cands], 40, eng
        )
        for slot, g in zip(v_all, got):
            slot[i] = g
        k = (7 * i) % n_cands                                           # sample vs the oracle
        assert got[k] == oracle.biprime_test_v_calculation(g_values[k], i, moduli[k], cands[k][0][i - 1], cands[k][1][i - 1], 40)
    verdicts = biprime.biprime_test_with_v_i_batch(v_all, moduli, 40, eng)
    for (pp, qp, m, want), verdict, vd in zip(cands, verdicts, v_all):
        if want:
            assert verdict is True
        else:
            assert verdict == oracle.biprime_test_with_v_i(vd, m, 40)
    assert sum(verdicts) == sum(1 for c in cands if c[3])
