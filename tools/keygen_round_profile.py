#!/usr/bin/env python3
"""Where the host time of one key-generation round goes (biprime.BiprimeRound, 65 536 candidates, key_length 2048, 5
parties): cProfile of the three compute steps, int in -> verdicts out.   usage: keygen_round_profile.py [batch_size]"""
import cProfile
import os
import pstats
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sympy
import torch

from protocols.distributed_keygen_amd import Engine, biprime, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
eng = Engine()
rng = random.Random(77)
n_parties, t, half = 5, 2, 1024
degree = 2 * t
prime = synthetic.random_prime(rng, 2 * (half + 4) + 44, mod4=1)
prime_list = [int(q) for q in sympy.primerange(3, 2001)]
shares = [synthetic.candidate_shares(rng, n_parties, half) for _ in range(B)]
mods = [sum(p) * sum(q) for p, q in shares]
points = list(range(1, n_parties + 1))
columns = {i: [] for i in points}
for m in mods:
    coeffs = [m] + [rng.getrandbits(prime.bit_length() + 8) % prime for _ in range(degree)]
    for i in points:
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * i + c) % prime
        columns[i].append(acc)


def one_round(profiles=None):
    """profiles: three cProfile.Profile objects, one per timed step (None: wall clock only)."""
    def step(k, fn):
        if profiles is None:
            return fn()
        profiles[k].enable()
        try:
            return fn()
        finally:
            profiles[k].disable()

    rnd = biprime.BiprimeRound(eng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(0, lambda: rnd.reconstruct_and_sieve(columns, prime, degree, prime_list, points=points))
    t1 = time.perf_counter()
    surv = rnd.survivors
    g_rng = random.Random(B)
    g_values = [[g_rng.getrandbits(2048 + 64) % m for _ in range(160)] for m in rnd.moduli]
    p1, q1 = [shares[k][0][0] for k in surv], [shares[k][1][0] for k in surv]
    t2 = time.perf_counter()
    v1 = step(1, lambda: rnd.v_calculation(g_values, 1, p1, q1, 40))
    t3 = time.perf_counter()
    v_by = [{1: v} for v in v1]
    for i in range(2, n_parties + 1):
        vi = biprime.biprime_test_v_calculation_batch(g_values, i, rnd.moduli, [shares[k][0][i - 1] for k in surv], [shares[k][1][i - 1] for k in surv], 40, eng)
        for d, v in zip(v_by, vi):
            d[i] = v
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    step(2, lambda: rnd.verdicts(v_by, 40, errors="return"))
    t5 = time.perf_counter()
    return t1 - t0, t3 - t2, t5 - t4, len(surv)


one_round()
best = None
for _ in range(3):
    cur = one_round()
    if best is None or sum(cur[:3]) < sum(best[:3]):
        best = cur
a, b, c, ns = best
print(f"batch {B}: {ns} survivors; reconstruct+sieve {a * 1e3:.1f} ms, v-calculation {b * 1e3:.1f} ms, verdicts {c * 1e3:.1f} ms = {B / (a + b + c) / 1e3:.0f} k candidates/s")
prs = [cProfile.Profile() for _ in range(3)]
one_round(prs)
for name, pr in zip(("reconstruct + sieve", "v-calculation (this party)", "verdicts"), prs):
    print(f"---- {name}")
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
