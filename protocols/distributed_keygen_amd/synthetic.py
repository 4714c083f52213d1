"""Deterministic synthetic inputs of the hot path's shapes: threshold-Paillier keys
with the structure distributed_keygen.py:1364-1500 produces, candidate moduli of the
distributed_keygen.py:855-876 shape, ciphertext batches.  Pure Python big-int; used by bench.py,
smoke() and the tests.  No reference code is involved.
"""

from __future__ import annotations

import math
import random
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

SEED = 0xD15C0

_SMALL_PRIMES = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47, 53, 59, 61, 67, 71, 73, 79, 83, 89, 97]


def is_probable_prime(n: int, rng: random.Random, rounds: int = 24) -> bool:
    if n < 2:
        return False
    for p in _SMALL_PRIMES:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for _ in range(rounds):
        a = rng.randrange(2, n - 1)
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def random_prime(rng: random.Random, bits: int, mod4: int = 3) -> int:
    while True:
        c = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        c += (mod4 - c) % 4
        if c.bit_length() == bits and is_probable_prime(c, rng):
            return c


@dataclass
class SharedKey:
    """What every party's PaillierSharedKey carries (paillier_shared_key.py:30-50), for all parties."""

    key_length: int
    n_parties: int
    t: int
    n: int
    p: int
    q: int
    n_fac: int
    degree: int
    theta: int
    shares: Dict[int, int] = field(default_factory=dict)

    @property
    def n_square(self) -> int:
        return self.n * self.n

    @property
    def theta_inv(self) -> int:
        return pow(self.theta, -1, self.n)

    def exponent(self, player_id: int) -> int:
        """Lagrange-folded partial-decryption exponent, paillier_shared_key.py:70-85."""
        others = [i + 1 for i in range(self.degree + 1) if i + 1 != player_id]
        num = math.prod(others)
        den = math.prod(j - player_id for j in others)
        return (self.n_fac * num * self.shares[player_id]) // den


def make_key(key_length: int, n_parties: int = 3, t: int = 1, kappa: int = 40, seed: int = SEED) -> SharedKey:
    """A key whose primes are the sums of `n_parties` additive shares of key_length/2 bits each
    (distributed_keygen.py:874-876), so N has key_length+2..key_length+5 bits as real keys do."""
    rng = random.Random(seed * 1000003 + key_length * 31 + n_parties * 7 + t)
    half = key_length // 2
    extra = max(1, (n_parties - 1).bit_length())
    lo = n_parties << (half - 1)
    while True:
        p = random_prime(rng, half + extra)
        q = random_prime(rng, half + extra)
        if p != q and p >= lo and q >= lo and p < (n_parties << half) and q < (n_parties << half):
            break
    n = p * q
    n_fac = math.factorial(n_parties)
    lam = n - p - q + 1
    beta = sum(rng.randrange(n) for _ in range(n_parties))
    bound = (n_fac**2) * (1 << kappa) * n * n_parties

    def poly(secret: int) -> List[int]:
        return [n_fac * secret] + [rng.randrange(-bound, bound) for _ in range(t)]

    def ev(coeffs: List[int], x: int) -> int:
        return sum(c * x**k for k, c in enumerate(coeffs))

    fl, fb = poly(lam), poly(beta)
    shares = {i: ev(fl, i) * ev(fb, i) for i in range(1, n_parties + 1)}
    theta = (lam * beta * n_fac**3) % n
    return SharedKey(key_length, n_parties, t, n, p, q, n_fac, 2 * t, theta, shares)


def random_ciphertexts(key: SharedKey, count: int, seed: int = SEED) -> List[int]:
    """Uniform residues modulo N^2.  Every unit of Z_{N^2} is (1+N)^m r^N for exactly one (m, r), so
    these are valid Paillier ciphertexts of uniformly random plaintexts."""
    rng = random.Random(seed ^ 0xC1F3)
    n2 = key.n_square
    nbytes = (n2.bit_length() + 7) // 8 + 8
    return [int.from_bytes(rng.randbytes(nbytes), "little") % n2 for _ in range(count)]


def encrypt(key: SharedKey, m: int, rng: random.Random) -> int:
    n, n2 = key.n, key.n_square
    r = rng.randrange(1, n)
    return (1 + m * n) % n2 * pow(r, n, n2) % n2


def candidate_shares(rng: random.Random, n_parties: int, prime_length: int) -> Tuple[List[int], List[int]]:
    """Additive shares p_i, q_i of the distributed_keygen.py:855-876 shape (party 1 is 3 mod 4)."""

    def one(index: int) -> int:
        return (1 << (prime_length - 1)) + (rng.getrandbits(prime_length - 3) << 2) + (3 if index == 1 else 0)

    return [one(i + 1) for i in range(n_parties)], [one(i + 1) for i in range(n_parties)]
