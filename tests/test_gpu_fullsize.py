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
    import hostpow          # tests/hostpow.py: mpz_powm on the host cores, checked against CPython pow per process

    i_pos = next(i for i in (1, 2, 3) if key.exponent(i) >= 0)
    idx = list(range(0, batch, 4))
    want = hostpow.powmod_many([(cts[k], key.exponent(i_pos), n2) for k in idx])
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


def _steer_to_primes(args):
    """(p_parts, q_parts, seed) -> the same shares with the last one of each moved so that both sums are prime."""
    from protocols.distributed_keygen_amd import synthetic

    p_parts, q_parts, seed = args
    rng = random.Random(seed)
    for parts in (p_parts, q_parts):
        base = t = sum(parts)
        while not synthetic.is_probable_prime(t, rng, 8):
            t += 4
        parts[-1] += t - base
    return p_parts, q_parts


def _host_v_row(args):
    """One candidate's reference v-calculation (DK:1084-1099) with CPython pow: (g list, N, exponent) -> v list."""
    gs, m, e = args
    kept = [g for g in gs if oracle.jacobi_symbol(g, m) == 1][:40]
    return [pow(g, e, m) for g in kept]


def test_c4_full_size_4096_candidates_key2048(eng):
    """configs[3] at its full size: 5 parties, key_length 2048, 4096 candidate moduli x 160 generators.  All five
    parties' v values through the fused Jacobi -> selection -> modexp path, the verdict of every candidate: exactly
    the planted biprimes pass all 40 slots.  72 candidates (the planted ones and a spread of composites) are
    recomputed on the host cores — Jacobi selection with the oracle, v values with pow() — and compared bit for
    bit, for party 1 (long exponent) and party 3 (short exponent).  The 512-candidate shard that one of 8 GPUs
    gets (another launch shape) reproduces its slice of the full run."""
    import multiprocessing as mp

    import torch

    from protocols.distributed_keygen_amd import limbs as L, synthetic

    n_parties, cands, gens, keep = 5, 4096, 160, 40
    rng = random.Random(0xC4)
    primes = oracle.small_prime_list(2000)
    shares = []                                       # survivors of the sieve DK:1288-1292, as the v-calculation sees them
    while len(shares) < cands:
        chunk = [synthetic.candidate_shares(rng, n_parties, 1024) for _ in range(16384)]
        bad = eng.sieve_batch([sum(p) * sum(q) for p, q in chunk], primes)
        shares += [sh for sh, b in zip(chunk, bad) if not b]
    shares = shares[:cands]
    planted = sorted(rng.sample(range(1, cands - 1), 8) + [0, cands - 1])
    with mp.Pool() as pool:
        fixed = pool.map(_steer_to_primes, [(shares[k][0], shares[k][1], 1000 + k) for k in planted], chunksize=1)
        for k, sh in zip(planted, fixed):
            shares[k] = sh
        mods = [sum(p) * sum(q) for p, q in shares]
        bits = max(m.bit_length() for m in mods)
        limbs = L.limbs_for_bits(bits)
        nb = (bits + 7) // 8 + 8
        g_all = [int.from_bytes(rng.randbytes(nb), "little") % m for m in mods for _ in range(gens)]
        g_t = eng.to_device(L.pack(g_all, limbs))
        mods_op = (eng.to_device(L.pack(mods, limbs)), bits)
        exps = {i: [((m - p[0] - q[0] + 1) // 4) if i == 1 else ((p[i - 1] + q[i - 1]) // 4) for m, (p, q) in zip(mods, shares)]
                for i in range(1, n_parties + 1)}
        v_all = torch.zeros((n_parties, cands, keep, limbs), dtype=torch.int32, device=eng.device)
        counts = {}
        for i in range(1, n_parties + 1):
            eb = max(e.bit_length() for e in exps[i])
            ex_op = (eng.to_device(L.pack(exps[i], L.limbs_for_bits(eb))), eb)
            v_t, cnt_t = eng.biprime_v_t(g_t, mods_op, ex_op, gens, keep)
            v_all[i - 1] = v_t.view(cands, keep, limbs)
            counts[i] = cnt_t.cpu().tolist()
            if i == 1:      # the shard one of 8 GPUs would get: its own launch (512 candidates), same rows
                sl = slice(3 * 512, 4 * 512)
                sh_mods = (mods_op[0][sl].contiguous(), bits)
                sh_exps = (ex_op[0][sl].contiguous(), eb)
                sv_t, scnt_t = eng.biprime_v_t(g_t[sl.start * gens : sl.stop * gens].contiguous(), sh_mods, sh_exps, gens, keep)
                assert torch.equal(sv_t.view(512, keep, limbs), v_all[0][sl]) and scnt_t.cpu().tolist() == counts[1][sl]
        assert all(counts[i] == counts[1] for i in counts)               # the selection does not depend on the party
        assert min(counts[1]) == keep                                    # 160 generators of a sieved candidate hold 40 with symbol 1
        passes = eng.biprime_verdict_t(v_all, mods_op).cpu().numpy()
        all_pass = [k for k in range(cands) if passes[k].all()]
        assert all_pass == planted                                       # the vote: planted biprimes, nothing else
        # a non-biprime fails a slot with probability >= 1/2 and in practice almost surely: none gets through 8 slots
        assert all(not passes[k][:8].all() for k in range(cands) if k not in planted)
        sample = sorted(set(planted + [(k * 577) % cands for k in range(62)]))
        assert len(sample) >= 64
        for party in (1, 3):
            want = pool.map(_host_v_row, [(g_all[k * gens : (k + 1) * gens], mods[k], exps[party][k]) for k in sample], chunksize=1)
            got = L.unpack(eng.to_host(v_all[party - 1][sample].reshape(-1, limbs)))
            assert [got[j * keep : (j + 1) * keep] for j in range(len(sample))] == want, party
