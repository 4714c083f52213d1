"""The oracle (oracle/oracle.py) against the golden vectors recorded from the reference itself.

Golden vectors: tests/golden/*.json, produced by tests/golden/make_golden.py by running the
unmodified reference modules (PSK:52-127, DK:1056-1209) in the build container.
"""

from __future__ import annotations

import random

import pytest
import sympy

from oracle import oracle
from conftest import unhex


def _decrypt_groups(*dicts):
    for d in dicts:
        for name, grp in d.items():
            yield name, grp


def test_partial_decrypt_and_combine_match_reference(golden_ref_keys, golden_decrypt_synth):
    checked = 0
    for name, grp in _decrypt_groups(golden_ref_keys, golden_decrypt_synth):
        n = unhex(grp["n"])
        degree, n_fac = grp["degree"], unhex(grp["n_fac"])
        theta_inv = unhex(grp["theta_inv"])
        assert oracle.mod_inv(unhex(grp["theta"]), n) == theta_inv
        shares = {int(i): unhex(v) for i, v in grp["shares"].items()}
        assert n_fac == oracle.n_factorial(grp["n_parties"])
        for case in grp["cases"]:
            c = unhex(case["c"])
            partials = {int(i): unhex(v) for i, v in case["partials"].items()}
            if "corrupt" not in name:
                for i, share in shares.items():
                    assert oracle.partial_decrypt(c, n, i, degree, n_fac, share) == partials[i], (name, i)
            if case["error"] == "ValueError":
                with pytest.raises(ValueError):
                    oracle.decrypt_combine(partials, n, degree, theta_inv)
            else:
                assert oracle.decrypt_combine(partials, n, degree, theta_inv) == unhex(case["m"])
            checked += 1
    assert checked >= 80


def test_negative_lagrange_exponent_present(golden_decrypt_synth):
    """PSK:89-91 (negative exponent → inverse) is exercised by the vectors."""
    grp = golden_decrypt_synth["k2048_n3_t1"]
    exps = [
        oracle.partial_decrypt_exponent(int(i), grp["degree"], unhex(grp["n_fac"]), unhex(s))
        for i, s in grp["shares"].items()
    ]
    assert any(e < 0 for e in exps) and any(e > 0 for e in exps)


def test_combine_missing_share_raises_keyerror(golden_decrypt_synth):
    grp = golden_decrypt_synth["k128_n3_t1"]
    case = grp["cases"][0]
    partials = {int(i): unhex(v) for i, v in case["partials"].items()}
    del partials[2]
    with pytest.raises(KeyError):
        oracle.decrypt_combine(partials, unhex(grp["n"]), grp["degree"], unhex(grp["theta_inv"]))


def test_biprime_v_and_verdict_match_reference(golden_biprime):
    for cand in golden_biprime["candidates"]:
        modulus = unhex(cand["modulus"])
        g_values = [unhex(g) for g in cand["g_values"]]
        nbip = cand["correct_param_biprime"]
        v_by_party = {}
        for i in range(1, cand["n_parties"] + 1):
            p_i, q_i = unhex(cand["p_parts"][i - 1]), unhex(cand["q_parts"][i - 1])
            got = oracle.biprime_test_v_calculation(g_values, i, modulus, p_i, q_i, nbip)
            assert got == [unhex(v) for v in cand["v"][str(i)]], (cand["label"], i)
            v_by_party[i] = got
        if cand["verdict"] == "KeyError":
            with pytest.raises(KeyError):
                oracle.biprime_test_with_v_i(v_by_party, modulus, nbip)
        else:
            assert oracle.biprime_test_with_v_i(v_by_party, modulus, nbip) is cand["verdict"], cand["label"]


def test_sieve_matches_reference(golden_biprime):
    for block in golden_biprime["sieve"]:
        primes = oracle.small_prime_list(block["prime_threshold"])
        assert len(primes) == block["n_primes"]
        assert primes[:3] == block["first"] and primes[-1] == block["last"]
        for case in block["cases"]:
            assert oracle.small_prime_divisors_test(primes, unhex(case["modulus"])) is case["has_small_divisor"]


def test_jacobi_matches_sympy():
    rng = random.Random(7)
    for bits in (8, 16, 64, 131, 1027):
        for _ in range(60):
            n = rng.getrandbits(bits) | 1
            if n < 3:
                continue
            a = rng.randrange(0, n)
            assert oracle.jacobi_symbol(a, n) == sympy.jacobi_symbol(a, n)
    assert oracle.jacobi_symbol(0, 9) == 0 and oracle.jacobi_symbol(0, 1) == 1


def test_small_prime_list_matches_sympy():
    for thr in (1, 2, 3, 10, 200, 2000):
        assert oracle.small_prime_list(thr) == [int(p) for p in sympy.primerange(3, thr + 1)]


def test_mult_list():
    assert oracle.mult_list([]) == 1
    assert oracle.mult_list([3, -4, 5]) == -60
    assert oracle.mult_list([3, 4, 5], 7) == 60 % 7


def test_shamir_field_steps_match_reference(golden_reconstruct):
    """DK:1274-1284 as the reference ran it (its own _generate_pq / Batched[ShamirVariable] flow over an
    in-memory pool): every party's share of every candidate modulus, and the reconstructed moduli."""
    for label, grp in golden_reconstruct.items():
        prime, degree = unhex(grp["prime"]), grp["degree"]
        shares = {int(i): {k: [unhex(v) for v in vals] for k, vals in d.items()} for i, d in grp["shares"].items()}
        count = len(grp["moduli"])
        for i, d in shares.items():
            assert [oracle.shamir_mul_add(d["p"][k], d["q"][k], d["zero"][k], prime) for k in range(count)] == d["n"], (label, i)
        for k in range(count):
            assert oracle.shamir_reconstruct({i: d["n"][k] for i, d in shares.items()}, prime, degree) == unhex(grp["moduli"][k])
        # pinned by the reference's own code alone (fixture `provenance`): N_k = (sum of the parties' additive p shares)
        # * (sum of their additive q shares), both sampled by DistributedPaillier._generate_pq itself (DK:854-876)
        p_add = [[unhex(v) for v in grp["p_additive"][i]] for i in sorted(grp["p_additive"])]
        q_add = [[unhex(v) for v in grp["q_additive"][i]] for i in sorted(grp["q_additive"])]
        assert [unhex(m) for m in grp["moduli"]] == [sum(p[k] for p in p_add) * sum(q[k] for q in q_add) for k in range(count)]
        assert all(p[k] % 4 == (3 if i == 0 else 0) for i, p in enumerate(p_add) for k in range(count))      # DK:855-858
        # the oracle's sharing -> multiply-add -> reconstruct pipeline from those additive shares lands on the same moduli
        import random as _random

        rng = _random.Random(7)
        n_parties, t = grp["n_parties"], grp["t"]
        for k in range(count):
            ps = oracle.shamir_share(sum(p[k] for p in p_add), prime, n_parties, t, rng)
            qs = oracle.shamir_share(sum(q[k] for q in q_add), prime, n_parties, t, rng)
            zs = oracle.shamir_share(0, prime, n_parties, 2 * t, rng)
            n_sh = {i: oracle.shamir_mul_add(ps[i], qs[i], zs[i], prime) for i in ps}
            assert oracle.shamir_reconstruct(n_sh, prime, degree) == unhex(grp["moduli"][k])
        # the moduli have the documented size: sum of n shares of key_length/2 bits, squared (SURVEY hard part 1)
        for m in grp["moduli"]:
            assert grp["key_length"] <= unhex(m).bit_length() <= grp["key_length"] + 2 * (grp["n_parties"] - 1).bit_length() + 1


def test_golden_partial_decryptions_agree_with_gmpy2(golden_decrypt_synth, golden_ref_keys, tmp_path):
    """BASELINE.json asks for results bit-exact against the gmpy2 reference: the recorded outputs of the
    reference (arithmetic leaf = CPython pow when they were generated) are recomputed here with
    gmpy2.powmod / gmpy2.invert — the leaf the reference uses when gmpy2 is installed — under the
    interpreter of this image that has gmpy2.  Skipped where that interpreter is absent."""
    import json
    import os
    import subprocess

    py = "/opt/conda/bin/python3.9"
    if not os.path.exists(py):
        pytest.skip("no interpreter with gmpy2 in this image")
    jobs = []
    for src in (golden_ref_keys, golden_decrypt_synth):
        for name, grp in src.items():
            if "corrupt" in name:
                continue
            n = unhex(grp["n"])
            for i, share in grp["shares"].items():
                exp = oracle.partial_decrypt_exponent(int(i), grp["degree"], unhex(grp["n_fac"]), unhex(share))
                for c in grp["cases"][:2]:
                    jobs.append({"c": c["c"], "e": hex(exp) if exp >= 0 else "-" + hex(-exp), "m": hex(n * n), "want": c["partials"][i]})
    path = tmp_path / "jobs.json"
    path.write_text(json.dumps(jobs))
    script = (
        "import json,sys,gmpy2\\n"
        "u=lambda s:-int(s[1:],16) if s.startswith('-') else int(s,16)\\n"
        "bad=0\\n"
        "for j in json.load(open(sys.argv[1])):\\n"
        "    c,e,m=gmpy2.mpz(u(j['c'])),u(j['e']),gmpy2.mpz(u(j['m']))\\n"
        "    if e<0: c,e=gmpy2.invert(c,m),-e\\n"
        "    bad+=int(gmpy2.powmod(c,e,m))!=u(j['want'])\\n"
        "print(bad,len(json.load(open(sys.argv[1]))))\\n"
    )
    r = subprocess.run([py, "-c", script.replace("\\n", "\n"), str(path)], capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "No module named" in r.stderr:
        pytest.skip("gmpy2 not importable")
    assert r.returncode == 0, r.stderr
    bad, total = map(int, r.stdout.split())
    assert bad == 0 and total == len(jobs) and total > 50


def test_precomputed_field_primes_of_the_gpu_suite_are_prime():
    """tests/test_gpu_ops.py takes its Shamir-field primes from a table (searching nextprime(2^4102) cost the GPU suite
    81 s): the small ones are sympy's nextprime exactly, the long ones prime (all the field kernels need)."""
    import importlib.util
    from pathlib import Path

    src = (Path(__file__).resolve().parent / "test_gpu_ops.py").read_text()
    line = next(l for l in src.splitlines() if l.startswith("NEXT_PRIME_OFFSET"))
    table = eval(line.split("=", 1)[1].split("#")[0])
    assert set(table) == {61, 133, 1030, 2054, 4102}
    for bits, off in table.items():
        p = (1 << bits) + off
        if bits <= 1030:
            assert p == sympy.nextprime(1 << bits), bits
        else:
            assert sympy.isprime(p), bits
