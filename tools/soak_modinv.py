#!/usr/bin/env python3
"""Randomised differential soak of the device inverse (csrc/mx_modinv.hpp: Kaliski's almost-inverse with multi-bit shifts and
word-wise halvings) against CPython pow(v, -1, m): odd moduli of 3 .. 16 000 bits incl. special ones (2^k - 1, 2^k + 1, squares,
moduli that fill their words), values incl. 1, m - 1, multiples of a factor (not invertible: ValueError expected), batches that
go through the product tree.   usage: soak_modinv.py [seed] [seconds]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocols.distributed_keygen_amd import Engine


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
    rng = random.Random(seed)
    eng = Engine()
    t0 = time.time()
    rounds = done = refused = 0
    while time.time() - t0 < budget:
        rounds += 1
        bits = rng.choice([rng.randint(3, 70), rng.randint(60, 2100), rng.randint(2000, 4200), rng.randint(4000, 8300), rng.randint(8000, 16000),
                           64, 128, 2048, 4096, 4102, 8192, 8198])
        kind = rng.random()
        if kind < 0.5:
            m = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        elif kind < 0.6:
            m = (1 << bits) - 1
        elif kind < 0.7:
            m = (1 << (bits - 1)) + 1
        elif kind < 0.85:
            h = rng.getrandbits(max(2, bits // 2)) | (1 << (max(2, bits // 2) - 1)) | 1
            m = h * h                                     # a square, as N^2
        else:
            f = rng.choice([3, 5, 7, 641, (1 << 61) - 1])
            m = f * (rng.getrandbits(max(2, bits - f.bit_length())) | 1)
        if m < 3:
            m = 3
        batch = rng.choice([1, 1, 1, 2, 3, 5, 17, 64])
        vals = [rng.choice([1, m - 1, 2, rng.randrange(1, m), rng.randrange(1, m), (m + 1) // 2]) for _ in range(batch)]
        if rng.random() < 0.15:
            from math import gcd
            g = next((p for p in (3, 5, 7, 641) if m % p == 0), None)
            if g:
                vals[rng.randrange(batch)] = g * rng.randrange(1, max(2, m // g))
        try:
            want = [pow(v, -1, m) for v in vals]
        except ValueError:
            want = None
        try:
            got = eng.modinv_batch(vals, m)
        except ValueError:
            got = None
        assert (got is None) == (want is None), (rounds, bits, batch, "invertibility")
        if want is None:
            refused += 1
        else:
            assert got == want, (rounds, bits, batch, [i for i, (x, y) in enumerate(zip(got, want)) if x != y][:4])
            done += batch
    print(f"soak_modinv seed {seed}: {rounds} rounds, {done} inverses bit-exact against pow(v, -1, m), {refused} batches refused like pow, in {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
