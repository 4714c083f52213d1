#!/usr/bin/env python3
"""CPU timing of ONE key-generation round the way the reference computes it (TEST/MEASUREMENT INFRASTRUCTURE — never
imported by the product; bench.py runs it as a subprocess for its `end_to_end_keygen.cpu_baseline`).

The reference executes a round sequentially on its asyncio thread (distributed_keygen.py:1284-1360):
    candidate_n.reconstruct()                          DK:1284   Lagrange interpolation per candidate (Python ints)
    __small_prime_divisors_test(prime_list, n)         DK:1197-1209, looped at :1288-1292   `%` per prime
    __biprime_test_v_calculation(g, index, n, ...)     DK:1056-1108, looped at :1313-1329   sympy.jacobi_symbol + pow_mod
    __biprime_test_with_v_i(...)                       DK:1110-1175, looped at :1339-1360   products and comparisons
with pow_mod = gmpy2.powmod when gmpy2 is installed.  This script times each of those four steps on a SAMPLE of a
round's candidates / survivors (job.json) on one core and reports seconds per unit; bench.py scales them to the
round size.  Runs under any Python >= 3.8 that has sympy; uses gmpy2 when importable (else CPython pow).

job.json: {"prime": hex, "points": [int], "columns": {point: [hex ...]}, "prime_list": [int], "moduli_check": [hex ...],
           "survivors": [{"modulus": hex, "exponent": hex, "g": [hex ...], "v_others": [[hex ...] per other party]}]}
"""

from __future__ import annotations

import json
import sys
import time

import sympy

try:
    import gmpy2  # type: ignore

    def pow_mod(b, e, m):
        return gmpy2.powmod(b, e, m)

    ENGINE = f"gmpy2 {gmpy2.version()} + sympy {sympy.__version__}"
except Exception:  # pragma: no cover
    pow_mod = pow
    ENGINE = f"CPython pow + sympy {sympy.__version__}"


def main() -> None:
    job = json.load(open(sys.argv[1]))
    prime = int(job["prime"], 16)
    points = job["points"]
    cols = {int(k): [int(v, 16) for v in vals] for k, vals in job["columns"].items()}
    prime_list = job["prime_list"]
    count = len(cols[points[0]])
    # ---- reconstruct: per candidate, the value at 0 of the polynomial through the shares (textbook Lagrange, as the
    # un-vendored ShamirShares.reconstruct_secret does it: coefficients recomputed per call)
    t0 = time.perf_counter()
    moduli = []
    for k in range(count):
        total = 0
        for i in points:
            num = den = 1
            for j in points:
                if j != i:
                    num = num * j % prime
                    den = den * (j - i) % prime
            total += cols[i][k] * num * pow(den, -1, prime)
        moduli.append(total % prime)
    t_rec = (time.perf_counter() - t0) / count
    assert [hex(m) for m in moduli[: len(job["moduli_check"])]] == job["moduli_check"], "reconstruction differs"
    # ---- sieve, DK:1197-1209
    t0 = time.perf_counter()
    bad = 0
    for n in moduli:
        for p in prime_list:
            if n % p == 0:
                bad += 1
                break
    t_sieve = (time.perf_counter() - t0) / count
    # ---- v calculation (DK:1084-1099) and verdict (DK:1147-1172) per survivor
    t_v = t_verdict = 0.0
    for sv in job["survivors"]:
        n, e = int(sv["modulus"], 16), int(sv["exponent"], 16)
        gs = [int(g, 16) for g in sv["g"]]
        others = [[int(v, 16) for v in row] for row in sv["v_others"]]
        t0 = time.perf_counter()
        vals = []
        for g in gs:
            if len(vals) >= 40:
                break
            if sympy.jacobi_symbol(g, n) == 1:
                vals.append(int(pow_mod(g, e, n)))
        t1 = time.perf_counter()
        ok = True
        for slot, v1 in enumerate(vals):
            product = 1
            for row in others:
                product *= row[slot]
            if v1 % n != product % n and v1 % n != (-product) % n:
                ok = False
                break
        t2 = time.perf_counter()
        t_v += t1 - t0
        t_verdict += t2 - t1
        assert [hex(v) for v in vals] == sv.get("v_check", [hex(v) for v in vals]), "v values differ"
    ns = max(1, len(job["survivors"]))
    print(json.dumps({"engine": ENGINE, "sample_candidates": count, "sample_survivors": len(job["survivors"]),
                      "reconstruct_s_per_candidate": t_rec, "sieve_s_per_candidate": t_sieve, "sieved_out_in_sample": bad,
                      "v_calculation_s_per_survivor": t_v / ns, "verdict_s_per_survivor": t_verdict / ns}))


if __name__ == "__main__":
    main()
