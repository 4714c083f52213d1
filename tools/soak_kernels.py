#!/usr/bin/env python3
"""Randomised differential soak of the round-2 kernels against CPython / sympy on all host cores:
Jacobi (divsteps + fallback, every NL instance, balanced / unbalanced / common-factor operands),
device inverse (every LPL instance), Shamir-field fma / lincomb.   soak_kernels.py [seed] [rounds]"""
import math
import multiprocessing as mp
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _jac(args):
    from sympy import jacobi_symbol

    return [int(jacobi_symbol(v, m)) for v, m in args]


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    from protocols.distributed_keygen_amd import Engine

    eng = Engine()
    rng = random.Random(seed)
    pool = mp.Pool()
    total = {"jacobi": 0, "modinv": 0, "field": 0}
    for rnd in range(rounds):
        # ---- Jacobi: groups of one modulus size each (a launch shares the limb count)
        for bits in (40, 95, 130, 200, 280, 500, 540, 1028, 1060, 2053, 2080, 3000, 4100):
            groups = 6 if bits < 3000 else 2
            mods, rows = [], []
            for g in range(groups):
                b = bits - rng.randrange(0, 33)
                m = rng.getrandbits(b) | (1 << (b - 1)) | 1
                if g == 1:
                    m = (3 * 5 * 7 * 11) * (rng.getrandbits(max(2, b - 12)) | 1)
                vals = []
                for k in range(64):
                    kind = k % 8
                    if kind == 0: v = rng.getrandbits(rng.randrange(1, 64))
                    elif kind == 1: v = m - rng.randrange(1, 1 << 20)
                    elif kind == 2: v = rng.getrandbits(max(1, b // 2))
                    elif kind == 3: v = (3 * 7) * rng.getrandbits(max(1, b - 8))
                    elif kind == 4: v = 1 << rng.randrange(0, b - 1)
                    else: v = rng.randrange(m)
                    vals.append(v % m)
                mods.append(m)
                rows.append(vals)
            got = eng.jacobi_batch(rows, mods)
            flat = [(v, m) for r, m in zip(rows, mods) for v in r]
            want_flat = [None] * len(flat)
            chunks = [flat[i::16] for i in range(16)]          # strided chunks over the host cores
            res = pool.map(_jac, chunks)
            for i, r in enumerate(res):
                want_flat[i::16] = r
            assert [x for r in got for x in r] == want_flat, ("jacobi", bits, rnd)
            total["jacobi"] += len(flat)
        # ---- modular inverse
        for bits in (30, 64, 700, 2051, 2100, 4102, 4200, 6100, 8198, 12000):
            m = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
            vals = [v for v in (rng.randrange(1, m) for _ in range(6)) if math.gcd(v, m) == 1][:4]
            assert eng.modinv_batch(vals, m) == [pow(v, -1, m) for v in vals], ("modinv", bits)
            big = [rng.randrange(1, m) for _ in range(70)]
            if all(math.gcd(v, m) == 1 for v in big) and bits < 5000:
                assert eng.modinv_batch(big, m) == [pow(v, -1, m) for v in big], ("modinv tree", bits)
            total["modinv"] += len(vals)
        # ---- Shamir field
        import sympy

        for bits in (70, 300, 1030, 2054):
            p = int(sympy.nextprime(rng.getrandbits(bits) | (1 << (bits - 1))))
            n = 257
            a, b, c = ([rng.randrange(p) for _ in range(n)] for _ in range(3))
            assert eng.shamir_fma_batch(a, b, c, p) == [(x * y + z) % p for x, y, z in zip(a, b, c)]
            terms = rng.randrange(1, 12)
            cols = [[rng.randrange(p) for _ in range(n)] for _ in range(terms)]
            cf = [rng.randrange(p) for _ in range(terms)]
            assert eng.shamir_lincomb_batch(cols, cf, p) == [sum(k * col[e] for k, col in zip(cf, cols)) % p for e in range(n)]
            total["field"] += 2 * n
        print("round", rnd, total, flush=True)
    print("soak ok", total)


if __name__ == "__main__":
    main()
