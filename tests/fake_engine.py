"""Test double for protocols.distributed_keygen_amd.engine.Engine (int-level API) built on the
oracle, so the HOST logic of the product (argument checks, ordering, error behaviour, batching,
sharding) is testable without a GPU.  Lives in tests/ only; the product never imports it."""

from __future__ import annotations

from typing import List, Sequence, Tuple

from oracle import oracle


class FakeEngine:
    def __init__(self) -> None:
        self.calls: List[Tuple[str, int]] = []

    def powmod_batch(self, bases: Sequence[int], exp: int, mod: int) -> List[int]:
        self.calls.append(("powmod_batch", len(bases)))
        if exp < 0:
            raise ValueError("negative exponent")
        return [oracle.pow_mod(b, exp, mod) for b in bases]

    def powmod_nsquare_batch(self, bases: Sequence[int], exp: int, n: int, keep_rows: bool = False):
        self.calls.append(("powmod_batch", len(bases)))
        if exp < 0:
            raise ValueError("negative exponent")
        out = [oracle.pow_mod(b, exp, n * n) for b in bases]
        return (out, list(out)) if keep_rows else out          # the "device column" of the double is the list itself

    def combine_columns(self, columns, n, theta_inv):
        from protocols.distributed_keygen_amd import codec

        batch = len(columns[0]) if columns else 0
        n2 = n * n
        cols = [[c % n2 for c in (codec.decode_int(v) for v in col)] for col in columns]
        return self.combine_batch([[col[k] for col in cols] for k in range(batch)], n, theta_inv)

    def powmod_batch_multi(self, bases, exps, mods):
        self.calls.append(("powmod_batch_multi", sum(len(b) for b in bases)))
        return [[oracle.pow_mod(b, e, m) for b in bs] for bs, e, m in zip(bases, exps, mods)]

    def modinv_batch(self, values, mod):
        self.calls.append(("modinv_batch", len(values)))
        return [oracle.mod_inv(v, mod) for v in values]

    def jacobi_batch(self, values, mods):
        self.calls.append(("jacobi_batch", sum(len(v) for v in values)))
        return [[oracle.jacobi_symbol(v, m) for v in vs] for vs, m in zip(values, mods)]

    def sieve_batch(self, candidates, primes):
        self.calls.append(("sieve_batch", len(candidates)))
        return [oracle.small_prime_divisors_test(primes, c) for c in candidates]

    def combine_batch(self, partials, n, theta_inv):
        self.calls.append(("combine_batch", len(partials)))
        msgs, ok = [], []
        for row in partials:
            try:
                msgs.append(oracle.decrypt_combine({i + 1: v for i, v in enumerate(row)}, n, len(row) - 1, theta_inv))
                ok.append(True)
            except ValueError:
                msgs.append(0)
                ok.append(False)
        return msgs, ok

    def biprime_verdict_batch(self, v, mods):
        self.calls.append(("biprime_verdict_batch", len(mods)))
        out = []
        for vc, m in zip(v, mods):
            row = []
            for k in range(len(vc[0])):
                prod = 1
                for i in range(1, len(vc)):
                    prod *= vc[i][k]
                row.append(vc[0][k] % m == prod % m or vc[0][k] % m == (-prod) % m)
            out.append(row)
        return out

    def shamir_fma_batch(self, a, b, c, prime):
        self.calls.append(("shamir_fma_batch", len(a)))
        return [oracle.shamir_mul_add(x, y, z, prime) for x, y, z in zip(a, b, c)]

    def shamir_lincomb_batch(self, columns, coeffs, prime):
        self.calls.append(("shamir_lincomb_batch", len(columns[0]) if columns else 0))
        return [sum(cf * col[e] for cf, col in zip(coeffs, columns)) % prime for e in range(len(columns[0]))]

    def shamir_reconstruct_sieve_batch(self, columns, coeffs, prime, primes):
        self.calls.append(("shamir_reconstruct_sieve_batch", len(columns[0]) if columns else 0))
        mods = [sum(cf * col[e] for cf, col in zip(coeffs, columns)) % prime for e in range(len(columns[0]))]
        bad = [oracle.small_prime_divisors_test(primes, m) for m in mods]
        return bad, {k: m for k, (m, b) in enumerate(zip(mods, bad)) if not b}
