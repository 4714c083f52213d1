"""GPU parity: sieve, share recombination and biprimality verdict vs the oracle / golden vectors."""

from __future__ import annotations

import random

import pytest

from conftest import unhex
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from protocols.distributed_keygen_amd import Engine

    return Engine()


# ------------------------------------------------------------------ sieve (DK:1197-1209)
def test_sieve_golden(eng, golden_biprime):
    for block in golden_biprime["sieve"]:
        primes = oracle.small_prime_list(block["prime_threshold"])
        cands = [unhex(c["modulus"]) for c in block["cases"]]
        assert eng.sieve_batch(cands, primes) == [c["has_small_divisor"] for c in block["cases"]]


@pytest.mark.parametrize("bits,threshold,batch", [(68, 200, 301), (1027, 2000, 1000), (2051, 2000, 777), (2051, 20000, 64), (8200, 2000, 9)])
def test_sieve_random(eng, bits, threshold, batch):
    rng = random.Random(bits + threshold)
    primes = oracle.small_prime_list(threshold)
    cands = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(batch)]
    cands[0] = primes[-1] * (cands[0] // primes[-1])          # divisible by the largest prime only if...
    cands[1] = 0
    cands[2] = 1
    cands[3] = primes[0]
    cands[4] = 2 ** (bits - 1)
    want = [oracle.small_prime_divisors_test(primes, c) for c in cands]
    assert eng.sieve_batch(cands, primes) == want
    assert 0 < sum(want) < batch


@pytest.mark.parametrize("bits,top", [(2051, (1 << 21) + 17), (2051, 1 << 26), (4102, 1 << 30), (8200, (1 << 31) - 1), (100, (1 << 31) - 1), (31, 1 << 22)])
def test_sieve_primes_beyond_two_to_the_21(eng, bits, top):
    """Lists with large primes (prime_threshold above 2^21: the reference takes any, DK:552-554): the 64-bit columns
    are folded every floor(2^32 / max prime) - 1 limbs — at 2^31 - 1 after every limb.  Candidates: multiples of each
    large prime alone, all-ones rows (the largest column sums), neighbours of multiples, random rows."""
    import sympy

    rng = random.Random(bits ^ top)
    large = [int(sympy.prevprime(top + 1))]
    while len(large) < 70:                                  # more than one 64-lane pass of primes
        large.append(int(sympy.prevprime(large[-1] - rng.randrange(1, 1 << 12))))
    primes = [3, 5, 7] + sorted(large)
    full = (1 << bits) - 1
    cands = []
    for q in large[:24]:
        m = (rng.getrandbits(bits) | (1 << (bits - 1))) // q * q
        while any(m % s == 0 for s in (3, 5, 7)) or m.bit_length() > bits:
            m -= q
        cands += [m, m + 2, m - q + 1]
    cands += [full, full - 1, q, q * q if (q * q).bit_length() <= bits else q, 1, 0]
    cands += [rng.getrandbits(bits) for _ in range(200)]
    cands = [c for c in cands if 0 <= c <= full]
    want = [oracle.small_prime_divisors_test(primes, c) for c in cands]
    assert eng.sieve_batch(cands, primes) == want
    assert sum(want) >= 24 and sum(want) < len(cands)
    with pytest.raises(Exception):
        eng.sieve_batch([15], [3, (1 << 31) + 11])


def test_sieve_edge(eng):
    assert eng.sieve_batch([], [3, 5]) == []
    assert eng.sieve_batch([15, 7], []) == [False, False]
    assert eng.sieve_batch([15, 7, 49], [3]) == [True, False, False]
    with pytest.raises(Exception):
        eng.sieve_batch([15], [2])


# ------------------------------------------------------------------ share recombination (PSK:95-127)
def test_combine_golden(eng, golden_ref_keys, golden_decrypt_synth):
    for src in (golden_ref_keys, golden_decrypt_synth):
        for name, grp in src.items():
            n, theta_inv, degree = unhex(grp["n"]), unhex(grp["theta_inv"]), grp["degree"]
            partials = [[unhex(c["partials"][str(i + 1)]) for i in range(degree + 1)] for c in grp["cases"]]
            msgs, ok = eng.combine_batch(partials, n, theta_inv)
            for case, m, good in zip(grp["cases"], msgs, ok):
                if case["error"] == "ValueError":
                    assert not good, name
                else:
                    assert good and m == unhex(case["m"]), name


@pytest.mark.parametrize("key", ["k128_n3_t1", "k1024_n3_t1", "k2048_n3_t1", "k2048_n5_t2"])
def test_combine_random_and_corrupted(eng, golden_decrypt_synth, key):
    grp = golden_decrypt_synth[key]
    n, theta_inv, degree, n_fac = unhex(grp["n"]), unhex(grp["theta_inv"]), grp["degree"], unhex(grp["n_fac"])
    shares = {int(i): unhex(s) for i, s in grp["shares"].items()}
    n2 = n * n
    rng = random.Random(hash(key) % 1000)
    base = [unhex(c["c"]) for c in grp["cases"]]
    partials, want = [], []
    for k in range(13):
        c = base[k % len(base)] * pow(base[(k + 1) % len(base)], k, n2) % n2     # homomorphic mixes
        ps = {i: oracle.partial_decrypt(c, n, i, degree, n_fac, shares[i]) for i in range(1, degree + 2)}
        if k % 4 == 3:
            ps[1 + k % (degree + 1)] = rng.randrange(n2)                          # inconsistent share
        partials.append([ps[i] for i in range(1, degree + 2)])
        try:
            want.append(oracle.decrypt_combine(ps, n, degree, theta_inv))
        except ValueError:
            want.append(None)
    # x == 0 and x == 1 corner cases of PSK:119-125
    partials.append([0] + [1] * degree)
    want.append(None)
    partials.append([1] * (degree + 1))
    want.append(0)
    msgs, ok = eng.combine_batch(partials, n, theta_inv)
    for m, good, w in zip(msgs, ok, want):
        assert (w is None and not good) or (good and m == w)
    assert any(w is None for w in want) and any(w is not None for w in want)


# ------------------------------------------------------------------ biprimality verdict (DK:1110-1175)
def test_verdict_golden(eng, golden_biprime):
    by_limbs = {}
    for cand in golden_biprime["candidates"]:
        if cand["verdict"] == "KeyError":
            continue
        nslots = min(len(v) for v in cand["v"].values())
        by_limbs.setdefault((unhex(cand["modulus"]).bit_length() // 40, cand["n_parties"], nslots), []).append(cand)
    for (_, npar, nslots), cands in by_limbs.items():
        mods = [unhex(c["modulus"]) for c in cands]
        v = [[[unhex(x) for x in c["v"][str(i)][:nslots]] for i in range(1, npar + 1)] for c in cands]
        got = eng.biprime_verdict_batch(v, mods)
        for c, slots, vv, m in zip(cands, got, v, mods):
            want_slots = []
            for k in range(nslots):
                prod = 1
                for i in range(1, npar):
                    prod *= vv[i][k]
                want_slots.append(vv[0][k] % m == prod % m or vv[0][k] % m == (-prod) % m)
            assert slots == want_slots, c["label"]
            full = nslots >= c["correct_param_biprime"]
            if c["verdict"] is True:
                assert full and all(slots)
            else:
                assert not all(slots) or not full


def test_verdict_negative_and_zero_products(eng):
    m = (1 << 200) + 235
    v1 = [5, m - 5, 0, 7, 1, m - 1]
    v2 = [5, 5, 0, 8, 1, 1]
    v3 = [1, 1, 9, 1, 1, 1]
    got = eng.biprime_verdict_batch([[v1, v2, v3]], [m])
    assert got == [[True, True, True, False, True, True]]


# ------------------------------------------------------------------ Jacobi symbol (DK:1089)
@pytest.mark.parametrize("bits,groups,gsize", [(20, 7, 33), (68, 9, 80), (131, 5, 160), (515, 3, 64), (1028, 4, 160), (2053, 3, 160), (4100, 2, 40),
                                               (4200, 2, 20), (8197, 2, 24)])
def test_jacobi_random(eng, bits, groups, gsize):
    rng = random.Random(bits)
    mods = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(groups)]
    vals = [[rng.randrange(m) for _ in range(gsize)] for m in mods]
    vals[0][0] = 0
    vals[0][1] = 1
    vals[0][2] = mods[0] - 1
    vals[0][3] = 2
    vals[1] = vals[1][: gsize - 5]                                   # ragged
    got = eng.jacobi_batch(vals, mods)
    assert got == [[oracle.jacobi_symbol(v, m) for v in vs] for vs, m in zip(vals, mods)]
    assert {x for row in got for x in row} >= {-1, 1}


def test_jacobi_special_cases(eng):
    assert eng.jacobi_batch([[0, 1, 2, 3, 4, 5, 6, 7, 8]], [9]) == [[oracle.jacobi_symbol(v, 9) for v in range(9)]]
    assert eng.jacobi_batch([[0, 5]], [1]) == [[1, 1]]
    m = (1 << 64) + 13                                             # common factor -> 0
    assert eng.jacobi_batch([[3 * 7, (m // 7) * 7 if m % 7 == 0 else 0]], [21 * 5 + 0 if False else 105]) == [[oracle.jacobi_symbol(21, 105), oracle.jacobi_symbol(0, 105)]]
    big = (1 << 2000) + 297
    pw = [1 << k for k in (1, 31, 32, 33, 64, 1999)]                # long runs of trailing zeros
    assert eng.jacobi_batch([pw], [big]) == [[oracle.jacobi_symbol(v, big) for v in pw]]
    with pytest.raises(ValueError):
        eng.jacobi_batch([[1]], [8])


# ------------------------------------------------------------------ mulmod / batched inverse / encryption
@pytest.mark.parametrize("bits,batch", [(136, 33), (2053, 40), (4102, 257), (8206, 9)])
def test_mulmod_and_modinv(eng, bits, batch):
    rng = random.Random(bits + batch)
    p, q = rng.getrandbits(bits // 2) | 1, rng.getrandbits(bits - bits // 2) | (1 << (bits - bits // 2 - 1)) | 1
    mod = p * q | 1
    a = [rng.randrange(mod) for _ in range(batch)]
    b = [rng.randrange(mod) for _ in range(batch)]
    a[0], b[0], a[1], b[1] = 0, 5, mod - 1, mod - 1
    assert eng.mulmod_batch(a, b, mod) == [x * y % mod for x, y in zip(a, b)]
    import math
    units = [v for v in b if math.gcd(v, mod) == 1]
    assert eng.modinv_batch(units, mod) == [pow(v, -1, mod) for v in units]
    assert eng.modinv_batch(units[:1], mod) == [pow(units[0], -1, mod)]
    with pytest.raises(ValueError):
        eng.modinv_batch(units[:3] + [0] + units[3:], mod)


def test_encrypt_batch_and_negative_exponent_party(eng, golden_decrypt_synth):
    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

    grp = golden_decrypt_synth["k2048_n3_t1"]
    n = unhex(grp["n"])
    rng = random.Random(8)
    msgs = [rng.randrange(n) for _ in range(20)]
    rs = [rng.randrange(1, n) for _ in range(20)]
    cts = eng.encrypt_batch(msgs, rs, n)
    assert cts == [(1 + m * n) % (n * n) * pow(r, n, n * n) % (n * n) for m, r in zip(msgs, rs)]
    keys = {
        int(i): GpuPaillierSharedKey(n, grp["t"], int(i), ShareView({int(i): unhex(s)}, grp["degree"], unhex(grp["n_fac"])),
                                     unhex(grp["theta"]), engine=eng)
        for i, s in grp["shares"].items()
    }
    assert any(k.lagrange_exponent() < 0 for k in keys.values())      # exercises the device inversion
    partials = {i: k.partial_decrypt_batch([PlainCiphertext(c, n) for c in cts]) for i, k in keys.items()}
    for i, k in keys.items():
        assert partials[i][3] == oracle.partial_decrypt(cts[3], n, i, grp["degree"], unhex(grp["n_fac"]), k.share.shares[i])
    assert keys[1].decrypt_batch([{i: partials[i][e] for i in keys} for e in range(20)]) == msgs


# ------------------------------------------------------------------ device-side selection + fused v-calculation
def test_select_first_and_fused_v_calculation(eng, golden_biprime):
    import numpy as np
    import torch

    from protocols.distributed_keygen_amd import limbs as L

    rng = random.Random(77)
    groups, gsize, keep, limbs = 5, 150, 40, 7
    rows = [rng.getrandbits(200) for _ in range(groups * gsize)]
    flags = np.array([rng.choice([-1, 0, 1, 1]) for _ in range(groups * gsize)], dtype=np.int8)
    flags[2 * gsize : 3 * gsize] = 0
    flags[2 * gsize + 149] = 1                               # a group with a single hit, in the last slot
    flags[3 * gsize : 3 * gsize + 64] = -1                   # first wave-pass empty
    out_t, cnt_t = eng.select_first_t(eng.to_device(L.pack(rows, limbs)), torch.from_numpy(flags).to(eng.device), gsize, keep)
    out = L.unpack(eng.to_host(out_t))
    cnt = cnt_t.cpu().numpy().tolist()
    for g in range(groups):
        want = [rows[g * gsize + k] for k in range(gsize) if flags[g * gsize + k] == 1][:keep]
        assert cnt[g] == len(want)
        assert out[g * keep : g * keep + keep] == want + [0] * (keep - len(want))
    # the fused device path reproduces the reference's v lists, including the short ones
    for cand in golden_biprime["candidates"]:
        modulus = unhex(cand["modulus"])
        gs = [unhex(g) for g in cand["g_values"]]
        for i in (1, 2):
            e = oracle.biprime_exponent(i, modulus, unhex(cand["p_parts"][i - 1]), unhex(cand["q_parts"][i - 1]))
            got = eng.biprime_v_batch([gs], [e], [modulus], cand["correct_param_biprime"])
            assert got == [[unhex(v) for v in cand["v"][str(i)]]], (cand["label"], i)


def test_jacobi_lanes_finishing_at_very_different_times(eng):
    """Regression: lanes whose numerator reaches 0 early must not disturb the wave-wide bookkeeping
    (live-limb tracking) of the lanes still running."""
    rng = random.Random(4)
    big = (1 << 2050) + 1234567
    big |= 1
    vals = []
    for k in range(192):
        vals.append([0, 1, 2, 3, 5, rng.getrandbits(20), rng.getrandbits(300), rng.randrange(big)][k % 8])
    mods = [big, (1 << 1500) + 7 | 1, (1 << 600) + 3 | 1]
    rows = [vals[:64], vals[64:128], vals[128:]]
    got = eng.jacobi_batch(rows, mods)
    assert got == [[oracle.jacobi_symbol(v, m) for v in r] for r, m in zip(rows, mods)]


def test_c_abi_from_plain_cpp(tmp_path):
    """examples/capi_partial_decrypt.cpp: the library used from C++/HIP without Python or PyTorch —
    mx_powmod_nsquare and mx_powmod_shared (two independent kernels) agree on 2048 partial decryptions
    at key_length 2048, and c^1 == c."""
    import shutil
    import subprocess
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not installed")
    pkg = root / "protocols" / "distributed_keygen_amd"
    exe = tmp_path / "capi_partial_decrypt"
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", f"-I{root / 'include'}", str(root / "examples" / "capi_partial_decrypt.cpp"),
                    f"-L{pkg}", "-lmxpaillier", f"-Wl,-rpath,{pkg}", "-o", str(exe)], check=True, capture_output=True)
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "identical through the one-shot form, the per-key plan" in run.stdout


# ------------------------------------------------------------------ Shamir field (DK:1274-1284)
def test_shamir_field_kernels_golden(eng, golden_reconstruct):
    """mx_fma_mod / mx_lincomb_mod against the values the reference itself produced in a keygen round."""
    from protocols.distributed_keygen_amd import shamir

    for label, grp in golden_reconstruct.items():
        prime, degree = unhex(grp["prime"]), grp["degree"]
        shares = {int(i): {k: [unhex(v) for v in vals] for k, vals in d.items()} for i, d in grp["shares"].items()}
        for i, d in shares.items():
            assert shamir.mul_add_shares_batch(d["p"], d["q"], d["zero"], prime, eng) == d["n"], (label, i)
        got = shamir.reconstruct_batch({i: d["n"] for i, d in shares.items()}, prime, degree, eng)
        assert got == [unhex(m) for m in grp["moduli"]]
        # the pin that owes nothing to the Shamir stand-in of make_golden.py (see `provenance` in the fixture): the
        # additive shares come from the reference's own candidate sampling, and N_k = (sum p_i[k]) * (sum q_i[k])
        p_add = [[unhex(v) for v in grp["p_additive"][i]] for i in sorted(grp["p_additive"])]
        q_add = [[unhex(v) for v in grp["q_additive"][i]] for i in sorted(grp["q_additive"])]
        assert got == [sum(p[k] for p in p_add) * sum(q[k] for q in q_add) for k in range(len(got))], label
        # ... and with a sharing the REFERENCE-side numbers alone determine up to the polynomials: this party shares its
        # own additive parts with fresh polynomials (the oracle's restatement of the un-vendored scheme), the device
        # multiplies, adds a sharing of zero and interpolates — the same moduli again
        rng = random.Random(len(label))
        n_parties, t, count = grp["n_parties"], grp["t"], len(got)
        ps = [oracle.shamir_share(sum(p[k] for p in p_add), prime, n_parties, t, rng) for k in range(count)]
        qs = [oracle.shamir_share(sum(q[k] for q in q_add), prime, n_parties, t, rng) for k in range(count)]
        zs = [oracle.shamir_share(0, prime, n_parties, 2 * t, rng) for k in range(count)]
        n_sh = {i: shamir.mul_add_shares_batch([s_[i] for s_ in ps], [s_[i] for s_ in qs], [s_[i] for s_ in zs], prime, eng)
                for i in range(1, n_parties + 1)}
        assert shamir.reconstruct_batch(n_sh, prime, degree, eng) == got, label


NEXT_PRIME_OFFSET = {61: 15, 133: 27, 1030: 603, 2054: 3133, 4102: 7065}      # nextprime(2^bits) - 2^bits


@pytest.mark.parametrize("bits,terms,batch", [(61, 3, 70), (133, 5, 33), (1030, 3, 129), (2054, 5, 64), (2054, 9, 17), (4102, 3, 9)])
def test_shamir_field_kernels_random(eng, bits, terms, batch):
    rng = random.Random(bits * 13 + terms)
    # sympy.nextprime(1 << bits), precomputed (the 4102-bit search alone took 81 s of the suite; the offsets are checked
    # against sympy on the CPU side, tests/test_oracle_golden.py)
    prime = (1 << bits) + NEXT_PRIME_OFFSET[bits]
    a = [rng.randrange(prime) for _ in range(batch)]
    b = [rng.randrange(prime) for _ in range(batch)]
    c = [rng.randrange(prime) for _ in range(batch)]
    a[0], b[0], c[0] = prime - 1, prime - 1, prime - 1
    a[1], b[1], c[1] = 0, 5, 0
    assert eng.shamir_fma_batch(a, b, c, prime) == [(x * y + z) % prime for x, y, z in zip(a, b, c)]
    cols = [[rng.randrange(prime) for _ in range(batch)] for _ in range(terms)]
    cols[0][0] = prime - 1
    coeffs = [rng.randrange(prime) for _ in range(terms)]
    coeffs[-1] = prime - 1
    want = [sum(cf * col[e] for cf, col in zip(coeffs, cols)) % prime for e in range(batch)]
    assert eng.shamir_lincomb_batch(cols, coeffs, prime) == want


def test_reconstructed_moduli_stay_on_the_device_for_the_sieve(eng, golden_reconstruct):
    """DK:1284-1292 without a host round trip: lincomb rows -> sieve_t."""
    from protocols.distributed_keygen_amd import limbs as L, shamir

    grp = golden_reconstruct["k1024_n3_t1"]
    prime, degree = unhex(grp["prime"]), grp["degree"]
    cols = [[unhex(v) for v in grp["shares"][str(i)]["n"]] for i in (1, 2, 3)]
    limbs = L.limbs_for(prime)
    import numpy as np

    x_t = eng.to_device(np.stack([L.pack(c, limbs) for c in cols]))
    mods_t = eng.shamir_lincomb_t(x_t, shamir.lagrange_coefficients_at_zero([1, 2, 3], prime), prime)
    primes = oracle.small_prime_list(2000)
    got = [bool(x) for x in eng.sieve_t(mods_t, primes).cpu().numpy()]
    assert got == [oracle.small_prime_divisors_test(primes, unhex(m)) for m in grp["moduli"]]


def test_combine_packed_rows_and_columns(eng, golden_decrypt_synth):
    """mx_combine_run with the status packed into the row (one all-gather unit), and the column-wise
    recombination the patched _decrypt_sequence_raw uses: device column + received ints + wire form."""
    import numpy as np
    import torch

    from protocols.distributed_keygen_amd import codec, limbs as L

    grp = golden_decrypt_synth["k1024_n3_t1"]
    n = unhex(grp["n"])
    n2 = n * n
    theta_inv = pow(unhex(grp["theta"]), -1, n)
    cases = grp["cases"]
    cols = [[unhex(c["partials"][str(i)]) for c in cases] for i in (1, 2, 3)]
    want = [unhex(c["m"]) for c in cases]
    limbs2, limbs = L.limbs_for(n2), L.limbs_for(n)
    bad_cols = [list(c) for c in cols]
    bad_cols[1][1] = (bad_cols[1][1] + 1) % n2                      # one inconsistent ciphertext
    pt = eng.to_device(np.stack([L.pack(c, limbs2) for c in bad_cols]))
    packed = eng.combine_t(pt, n, theta_inv, packed=True)
    assert tuple(packed.shape) == (len(cases), limbs + 1)
    msg_t, st_t = eng.combine_t(pt, n, theta_inv)
    assert torch.equal(packed[:, :limbs], msg_t) and packed[:, limbs].tolist() == [int(x) for x in st_t.tolist()]
    assert st_t.tolist() == [0, 1] + [0] * (len(cases) - 2)
    # columns: own partials kept on the device, one received as ints, one in wire form
    e = oracle.partial_decrypt_exponent(1, grp["degree"], unhex(grp["n_fac"]), unhex(grp["shares"]["1"]))
    cs = [unhex(c["c"]) for c in cases]
    bases = cs if e >= 0 else [oracle.mod_inv(c, n2) for c in cs]
    own, own_col = eng.powmod_nsquare_batch(bases, abs(e), n, keep_rows=True)
    assert own == cols[0] and hasattr(own_col, "data_ptr")
    msgs, ok = eng.combine_columns([own_col, cols[1], [codec.encode_int(v) for v in cols[2]]], n, theta_inv)
    assert all(ok) and msgs == want
    msgs, ok = eng.combine_columns([own_col, [v + 5 * n2 for v in cols[1]], [v - n2 for v in cols[2]]], n, theta_inv)
    assert all(ok) and msgs == want                                  # un-reduced / negative representatives


def test_chunked_int_level_batch_on_several_streams(eng):
    """Engine._pipelined: an int-level batch above PIPELINE_MIN runs as chunks on side streams with
    pinned staging; results, order and the kept device column equal the single-launch path."""
    from protocols.distributed_keygen_amd import limbs as L

    rng = random.Random(31337)
    n = rng.getrandbits(515) | (1 << 514) | 1
    n2 = n * n
    e = rng.getrandbits(130) | 1
    count = eng.PIPELINE_MIN + 1237
    bases = [rng.randrange(n2) for _ in range(count)]
    got, col = eng.powmod_nsquare_batch(bases, e, n, keep_rows=True)
    assert eng.last_timing["chunks"] >= 4
    idx = [0, 1, count // 2, count - 2, count - 1] + [rng.randrange(count) for _ in range(40)]
    assert [got[k] for k in idx] == [pow(bases[k], e, n2) for k in idx]
    assert L.unpack(eng.to_host(col)) == got
    small = eng.powmod_nsquare_batch(bases[:5000], e, n)
    assert small == got[:5000] and eng.last_timing["chunks"] == 1
    # a second chunked call reuses streams, workspaces and pinned buffers
    again = eng.powmod_nsquare_batch(bases, e, n)
    assert again == got


# ------------------------------------------------------------------ device modular inverse (PSK:50, PSK:89-91)
@pytest.mark.parametrize("bits", [5, 64, 65, 1027, 2051, 2080, 4102, 6200, 8198, 16700])
def test_modinv_direct_wave_kernel(eng, bits):
    """mx_modinv (one wavefront per element, every lanes-per-limb instance) vs pow(v, -1, m)."""
    import numpy as np

    from protocols.distributed_keygen_amd import limbs as L

    rng = random.Random(bits)
    mod = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
    limbs = L.limbs_for(mod)
    vals = [1, 2, mod - 1, mod - 2, (mod + 1) // 2] + [rng.randrange(1, mod) for _ in range(3)]
    import math

    vals = [v for v in vals if math.gcd(v, mod) == 1][:4]
    got = L.unpack(eng.to_host(eng.modinv_direct_t(eng.to_device(L.pack(vals, limbs)), mod)))
    assert got == [pow(v, -1, mod) for v in vals]


def test_modinv_not_invertible_raises_like_pow(eng):
    from protocols.distributed_keygen_amd import limbs as L

    p, q = (1 << 89) - 1, (1 << 107) - 1
    mod = p * q
    for bad in (0, p, 3 * q, mod - p):
        with pytest.raises(ValueError):
            eng.modinv_batch([5, bad, 7], mod)
    big = [(k * 7919 + 1) % mod for k in range(1, 40)]
    big[17] = 5 * p                                         # one non-invertible element poisons the tree root
    with pytest.raises(ValueError):
        eng.modinv_batch(big, mod)
    big[17] = 11
    assert eng.modinv_batch(big, mod) == [pow(v, -1, mod) for v in big]


def test_theta_inv_and_negative_exponent_have_no_host_inverse(eng, golden_decrypt_synth, monkeypatch):
    """PSK:50 and PSK:89-91 through the device inverse: builtins.pow with a negative exponent is
    forbidden for the duration of key construction and of a partial decryption with a negative
    Lagrange exponent; results equal the reference's recorded partial decryptions."""
    import builtins

    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

    real_pow = builtins.pow

    def guarded_pow(b, e, *m):
        assert not (m and isinstance(e, int) and e < 0), "host modular inverse on the product path"
        return real_pow(b, e, *m)

    grp = next(g for name, g in golden_decrypt_synth.items() if "corrupt" not in name and g["key_length"] >= 1024 and any(
        oracle.partial_decrypt_exponent(int(i), g["degree"], unhex(g["n_fac"]), unhex(s)) < 0 for i, s in g["shares"].items()))
    n = unhex(grp["n"])
    neg = next(int(i) for i, s in grp["shares"].items()
               if oracle.partial_decrypt_exponent(int(i), grp["degree"], unhex(grp["n_fac"]), unhex(s)) < 0)
    monkeypatch.setattr(builtins, "pow", guarded_pow)
    share = ShareView({neg: unhex(grp["shares"][str(neg)])}, grp["degree"], unhex(grp["n_fac"]))
    key = GpuPaillierSharedKey(n, grp["t"], neg, share, unhex(grp["theta"]), engine=eng)
    got = key.partial_decrypt_batch([PlainCiphertext(unhex(c["c"]), n) for c in grp["cases"]])
    monkeypatch.undo()
    assert key.theta_inv == pow(unhex(grp["theta"]), -1, n)
    assert got == [unhex(c["partials"][str(neg)]) for c in grp["cases"]]


def test_biprime_v_two_phase_jacobi_selection(eng):
    """The fused v-calculation evaluates the head of the generator list first and the tail only for
    candidates whose head did not yield `keep` generators with symbol 1 (DK:1084-1099): candidates
    built so that the 40th such generator lies in the head, exactly at the head/tail boundary, deep in
    the tail, or does not exist — all must equal the reference's sequential selection."""
    from protocols.distributed_keygen_amd import biprime

    rng = random.Random(160)
    mods, gens = [], []
    for c in range(12):
        m = rng.getrandbits(1027) | (1 << 1026) | 1
        ones = [g for g in (rng.randrange(2, m) for _ in range(900)) if oracle.jacobi_symbol(g, m) == 1]
        others = [g for g in (rng.randrange(2, m) for _ in range(600)) if oracle.jacobi_symbol(g, m) != 1]
        # number of symbol-1 generators placed in the first 104 positions
        in_head = [104, 60, 40, 39, 38, 20, 5, 0, 41, 39, 1, 0][c]
        in_tail = [0, 0, 0, 1, 2, 30, 56, 56, 0, 0, 38, 10][c]
        head = ones[:in_head] + others[: 104 - in_head]
        rng.shuffle(head)
        tail = ones[in_head : in_head + in_tail] + others[104 - in_head : 104 - in_head + 56 - in_tail]
        rng.shuffle(tail)
        mods.append(m)
        gens.append(head + tail)
        assert len(gens[-1]) == 160
    p = [rng.getrandbits(510) for _ in mods]
    q = [rng.getrandbits(510) for _ in mods]
    # (a launch this small evaluates all 160 symbols at once — a second launch would only add its latency —, so the
    # two-phase path is forced for one pass: both must give the reference's selection)
    for one_launch_below in (0, type(eng).JACOBI_ONE_LAUNCH_SYMBOLS):
        eng.JACOBI_ONE_LAUNCH_SYMBOLS = one_launch_below
        try:
            for index in (1, 2):
                got = biprime.biprime_test_v_calculation_batch(gens, index, mods, p, q, 40, eng)
                want = [oracle.biprime_test_v_calculation(g, index, m, pi, qi, 40) for g, m, pi, qi in zip(gens, mods, p, q)]
                assert got == want, (one_launch_below, index)
        finally:
            del eng.JACOBI_ONE_LAUNCH_SYMBOLS
    assert [len(w) for w in want] == [40, 40, 40, 40, 40, 40, 40, 40, 40, 39, 39, 10]
    # a short generator list (no tail at all) and keep larger than the list
    got = biprime.biprime_test_v_calculation_batch([g[:30] for g in gens], 1, mods, p, q, 40, eng)
    assert got == [oracle.biprime_test_v_calculation(g[:30], 1, m, pi, qi, 40) for g, m, pi, qi in zip(gens, mods, p, q)]


def test_jacobi_unbalanced_operands_and_safety_net(eng, monkeypatch):
    """Tiny numerators against large moduli, numerators just below the modulus, common factors — and
    the same inputs with the divstep batches cut short (mx_debug_knob), so that the plain
    binary algorithm that backs them up has to finish every symbol from an intermediate state."""
    rng = random.Random(77)
    rows, mods = [], []
    for bits in (61, 300, 1028, 2053, 4100, 8197):
        m = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        vals = [0, 1, 2, 3, 4, 5, 12345, m - 1, m - 2, m - 12345, (m + 1) // 2, rng.getrandbits(20), rng.getrandbits(bits // 3),
                3 * 5 * 7 * rng.getrandbits(40)] + [rng.randrange(m) for _ in range(18)]
        rows.append([v % m for v in vals])
        mods.append(m)
    mods.append(3 * 5 * 7 * 11 * 13 * ((1 << 500) + 1))
    rows.append([3, 5, 15, 1001, 17, (1 << 400) + 1] + [rng.randrange(mods[-1]) for _ in range(26)])
    want = [[oracle.jacobi_symbol(v, m) for v in r] for r, m in zip(rows, mods)]
    assert eng.jacobi_batch(rows, mods) == want
    try:
        for cut in (0, 1, 7, 60):
            eng.debug_knob("jacobi_max_batches", cut + 1)        # at most `cut` divstep batches
            assert eng.jacobi_batch(rows, mods) == want, cut
    finally:
        eng.debug_knob("jacobi_max_batches", 0)


def test_randomize_batch_keeps_the_plaintext(eng):
    """c * r^N mod N^2: bit-exact vs pow, and the re-randomised ciphertexts decrypt to the same messages."""
    from protocols.distributed_keygen_amd import synthetic

    key = synthetic.make_key(1024, 3, 1)
    n, n2 = key.n, key.n_square
    rng = random.Random(404)
    msgs = [0, 1, n - 1] + [rng.randrange(n) for _ in range(13)]
    cts = eng.encrypt_batch(msgs, [rng.randrange(1, n) for _ in msgs], n)
    rs = [rng.randrange(1, n) for _ in msgs]
    fresh = eng.randomize_batch(cts, rs, n)
    assert fresh == [c * pow(r, n, n2) % n2 for c, r in zip(cts, rs)] and fresh != cts
    partials = []
    for i in (1, 2, 3):
        e = key.exponent(i)
        bases = fresh if e >= 0 else eng.modinv_batch(fresh, n2)
        partials.append(eng.powmod_nsquare_batch(bases, abs(e), n))
    got, ok = eng.combine_columns(partials, n, key.theta_inv)
    assert all(ok) and got == msgs


def test_operator_surface_pow_mod_and_mod_inv(eng):
    """operators.pow_mod / mod_inv (the names the reference imports at DK:35, PSK:20) and their batched forms."""
    from protocols.distributed_keygen_amd import operators

    rng = random.Random(35)
    mod = rng.getrandbits(1027) | (1 << 1026) | 1
    vals = [rng.randrange(1, mod) for _ in range(9)]
    e = rng.getrandbits(300)
    assert operators.pow_mod(vals[0], e, mod, engine=eng) == pow(vals[0], e, mod)
    assert operators.pow_mod_batch(vals, e, mod, engine=eng) == [pow(v, e, mod) for v in vals]
    import math

    inv_ok = [v for v in vals if math.gcd(v, mod) == 1]
    assert operators.pow_mod_batch(inv_ok, -e, mod, engine=eng) == [pow(v, -e, mod) for v in inv_ok]
    assert operators.mod_inv(inv_ok[0], mod, engine=eng) == pow(inv_ok[0], -1, mod)
    with pytest.raises(ValueError):
        operators.mod_inv(0, mod, engine=eng)
    with pytest.raises(ValueError):
        operators.pow_mod(3, 5, 1 << 64, engine=eng)           # even modulus: no Montgomery arithmetic, no CPU path


def test_small_kernels_on_the_priority_companion_stream(eng, golden_decrypt_synth):
    """Engine.set_priority_aux: recombination, verdict and the Jacobi filter + selection run on a high-priority
    companion of the calling stream, ordered by events on both sides — same results, from several caller streams
    at once and without any host synchronisation in between."""
    import torch

    from protocols.distributed_keygen_amd import limbs as L

    grp = golden_decrypt_synth["k1024_n3_t1"]
    n, theta_inv = unhex(grp["n"]), unhex(grp["theta_inv"])
    n2 = n * n
    limbs2 = L.limbs_for(n2)
    cases = grp["cases"]
    partials = torch.stack([eng.to_device(L.pack([unhex(c["partials"][str(i)]) for c in cases], limbs2)) for i in (1, 2, 3)])
    want = [unhex(c["m"]) for c in cases]
    rng = random.Random(31)
    mods = [rng.getrandbits(515) | (1 << 514) | 1 for _ in range(6)]
    exps = [rng.getrandbits(500) for _ in mods]
    gens = [[rng.randrange(m) for _ in range(24)] for m in mods]
    eng.set_priority_aux(True)
    try:
        streams = [torch.cuda.Stream() for _ in range(3)]
        outs = []
        for k in range(6):
            with torch.cuda.stream(streams[k % 3]):
                m_t, st_t = eng.combine_t(partials, n, theta_inv)
                outs.append((m_t, st_t))
        torch.cuda.synchronize()
        for m_t, st_t in outs:
            assert L.unpack(eng.to_host(m_t)) == want and int(st_t.sum().item()) == 0
        got = eng.biprime_v_batch(gens, exps, mods, 8)
        for g, e, m, row in zip(gens, exps, mods, got):
            kept = [x for x in g if oracle.jacobi_symbol(x, m) == 1][:8]
            assert row == [pow(x, e, m) for x in kept]
        assert len(eng._aux) >= 3
    finally:
        eng.set_priority_aux(False)


def test_stream_concurrency_probe_and_clock_probe(eng):
    """mx_spin / mx_clock_probe: the streams the engine would cut a long batch over are verified to run side
    by side (at least one does; with 16 hardware queues all of them), and the shader clock measured by a probe
    wavefront is a plausible MI355X clock."""
    import torch

    streams = [torch.cuda.Stream(priority=-1) for _ in range(4)]
    ok = eng.stream_concurrency(streams, spin_us=300)
    assert 1 <= len(ok) <= 4 and ok[0] is streams[0]
    chunk = eng._chunk_streams(4)
    assert 1 <= len(chunk) <= 4
    mhz = eng.clock_probe_mhz(eng.clock_probe_start(200))
    assert 500.0 < mhz < 3000.0


def test_combine_columns_with_unreduced_peer_values(eng, golden_decrypt_synth):
    """a peer's partial decryptions that are not canonical residues (v + k N^2, as plain
    ints and in wire form) recombine to the same plaintexts."""
    from protocols.distributed_keygen_amd import codec

    grp = golden_decrypt_synth["k1024_n3_t1"]
    n, theta_inv = unhex(grp["n"]), unhex(grp["theta_inv"])
    n2 = n * n
    cols = [[unhex(c["partials"][str(i)]) for c in grp["cases"]] for i in (1, 2, 3)]
    want = [unhex(c["m"]) for c in grp["cases"]]
    cols[1] = [v + n2 for v in cols[1]]
    cols[2] = [codec.encode_int(v + 2 * n2) if k % 2 else v + n2 for k, v in enumerate(cols[2])]
    msgs, ok = eng.combine_columns(cols, n, theta_inv)
    assert all(ok) and msgs == want


def test_cu_slice_streams_run_small_launches_side_by_side(eng):
    """Engine.cu_slice_streams: launches on streams confined to disjoint slices of the CUs give the same results
    as on ordinary streams (four small N^2 modexp launches in flight, no host synchronisation in between)."""
    import torch

    from protocols.distributed_keygen_amd import limbs as L

    rng = random.Random(4242)
    n = rng.getrandbits(1027) | (1 << 1026) | 1
    n2 = n * n
    e = rng.getrandbits(500) | 1
    bases = [rng.randrange(n2) for _ in range(200)]
    rows = eng.to_device(L.pack(bases, L.limbs_for(n2)))
    streams = eng.cu_slice_streams(4)
    assert len(streams) == 4 and len({int(s.cuda_stream) for s in streams}) == 4
    assert eng.cu_slice_streams(4) is streams                       # created once per engine
    torch.cuda.synchronize()
    outs = []
    for k in range(8):
        with torch.cuda.stream(streams[k % 4]):
            outs.append(eng.powmod_nsquare_t(rows, n, e))
    torch.cuda.synchronize()
    want = [pow(b, e, n2) for b in bases]
    for out in outs:
        assert L.unpack(eng.to_host(out)) == want


def test_latency_geometry_setting_falls_back_for_wide_generic_moduli(eng):
    """an engine tuned with set_limbs_per_lane(3) for low-latency N^2 decryptions must still take
    generic moduli wider than the 3-limb generic instances exist for (5533 bits) — N^2 of key_length 4096 through
    powmod_batch / powmod_batch_multi — by leaving those launches to the automatic geometry, and must run the latency
    instances where they exist."""
    rng = random.Random(8200)
    eng.set_limbs_per_lane(3)
    try:
        mod = rng.getrandbits(8200) | (1 << 8199) | 1
        exp = rng.getrandbits(300)
        bases = [rng.randrange(mod) for _ in range(5)]
        assert eng.geometry(8200, 5, 1)[1] in (9, 18)
        assert eng.powmod_batch(bases, exp, mod) == [pow(b, exp, mod) for b in bases]
        mods = [rng.getrandbits(5600) | (1 << 5599) | 1 for _ in range(2)]
        exps = [rng.getrandbits(200) for _ in mods]
        groups = [[rng.randrange(m) for _ in range(3)] for m in mods]
        assert eng.powmod_batch_multi(groups, exps, mods) == [[pow(b, e, m) for b in bs] for bs, e, m in zip(groups, exps, mods)]
        small = rng.getrandbits(2053) | (1 << 2052) | 1
        assert eng.geometry(2053, 5, 1)[1] == 3
        assert eng.powmod_batch(bases[:3], exp, small) == [pow(b % small, exp, small) for b in bases[:3]]
    finally:
        eng.set_limbs_per_lane(0)


@pytest.mark.gpu
def test_round_kernels_with_moduli_of_very_different_lengths_in_one_launch(eng):
    """One launch has one geometry — taken from its longest modulus — and every candidate its own modulus: the Jacobi
    filter, the selection, the v modexps (every lane geometry incl. the bipartite form) and the verdict for candidates
    from full length down to a few bits in ONE call, against the oracle.  (A keygen round's candidates differ by a bit
    or two; round 5's soak found a group that came out as N + 1 when they differ by five, csrc/mx_bimont.hpp.)"""
    from oracle import oracle
    from protocols.distributed_keygen_amd import biprime

    rng = random.Random(55)
    n_parties, keep = 3, 6
    for bits in (1029, 2053):
        lens = [bits, bits - 1, bits - 6, bits - 33, bits * 2 // 3, bits // 2 + 2, bits // 3, 190, 64, 9]
        mods, shares = [], []
        for b in lens:
            while True:
                p = [rng.getrandbits(max(2, b // 2 - 2)) for _ in range(n_parties)]
                q = [rng.getrandbits(max(2, b // 2 - 2)) for _ in range(n_parties)]
                p[0] |= 3; q[0] |= 3
                for i in range(1, n_parties):
                    p[i] &= ~3; q[i] &= ~3
                m = sum(p) * sum(q)
                if m % 2 == 1 and m >= 9:
                    break
            mods.append(m); shares.append((p, q))
        gens = [[rng.randrange(m) for _ in range(24)] for m in mods]
        want_v = {i: [oracle.biprime_test_v_calculation(g, i, m, sh[0][i - 1], sh[1][i - 1], keep) for g, m, sh in zip(gens, mods, shares)]
                  for i in range(1, n_parties + 1)}
        try:
            for lpl in (0, 3, 6, 9, 18):
                eng.set_limbs_per_lane(lpl)
                for i in (1, 2):
                    got = biprime.biprime_test_v_calculation_batch(gens, i, mods, [sh[0][i - 1] for sh in shares], [sh[1][i - 1] for sh in shares], keep, eng)
                    assert got == want_v[i], (bits, lpl, i)
        finally:
            eng.set_limbs_per_lane(0)
        assert eng.jacobi_batch(gens, mods) == [[oracle.jacobi_symbol(g, m) for g in gs] for gs, m in zip(gens, mods)]
        v_by = [{i: want_v[i][c] for i in want_v} for c in range(len(mods))]
        want = []
        for c, m in enumerate(mods):
            try:
                want.append(oracle.biprime_test_with_v_i(v_by[c], m, keep))
            except KeyError as exc:
                want.append(type(exc))
        got = biprime.biprime_test_with_v_i_batch(v_by, mods, keep, eng, errors="return")
        assert [type(g) if isinstance(g, Exception) else g for g in got] == want, bits
