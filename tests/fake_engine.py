"""Test double for protocols.distributed_keygen_amd.engine.Engine (int-level API) built on the
oracle, so the HOST logic of the product (argument checks, ordering, error behaviour, batching,
sharding) is testable without a GPU.  Lives in tests/ only; the product never imports it."""

from __future__ import annotations

from typing import List, Sequence, Tuple

from oracle import oracle


class _Handle(tuple):
    """The double's "device handle": a tagged tuple (kind, payload) with the two slicing methods the product code uses on
    the real handles (engine._ModulusRows.repeated, engine._VRows.part)."""

    def __new__(cls, kind, payload):
        return super().__new__(cls, (kind, payload))

    def repeated(self, times):
        return _Handle(self[0], list(self[1]) * times)

    def part(self, k, parts):
        n = len(self[1]) // parts
        return _Handle(self[0], self[1][k * n:(k + 1) * n])


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

    def shamir_reconstruct_sieve_batch(self, columns, coeffs, prime, primes, keep_rows=False):
        self.calls.append(("shamir_reconstruct_sieve_batch", len(columns[0]) if columns else 0))
        mods = [sum(cf * col[e] for cf, col in zip(coeffs, columns)) % prime for e in range(len(columns[0]))]
        bad = [oracle.small_prime_divisors_test(primes, m) for m in mods]
        surviving = {k: m for k, (m, b) in enumerate(zip(mods, bad)) if not b}
        if keep_rows:          # the double's "device rows" of the survivors' moduli: a tagged list
            return bad, surviving, _Handle("mods_rows", [surviving[k] for k in sorted(surviving)]) if surviving else None
        return bad, surviving

    # the device-resident forms of a key-generation round (biprime.BiprimeRound); handles are tagged Python lists
    def biprime_v_batch(self, g_values, exps, mods, keep, mods_rows=None, keep_rows=False):
        self.calls.append(("biprime_v_batch", len(mods)))
        if mods_rows is not None:
            assert mods_rows[0] == "mods_rows" and mods_rows[1] == list(mods), "kept moduli rows of other candidates"
        out = []
        for gs, e, m in zip(g_values, exps, mods):
            sel = [g for g in gs if oracle.jacobi_symbol(g, m) == 1][:keep]
            out.append([oracle.pow_mod(g, e, m) for g in sel])
        return (out, _Handle("v_rows", [list(v) for v in out])) if keep_rows else out

    def biprime_verdict_columns(self, columns, mods, n_slots, mods_rows=None):
        self.calls.append(("biprime_verdict_columns", len(mods)))
        if mods_rows is not None:
            assert mods_rows[0] == "mods_rows" and mods_rows[1] == list(mods)
        cols = []
        for col in columns:
            if isinstance(col, tuple) and col[0] == "v_rows":       # this party's values "from the device"
                self.calls.append(("own_column_from_device", len(col[1])))
                cols.append([x for v in col[1] for x in (list(v[:n_slots]) + [0] * (n_slots - min(n_slots, len(v))))])
            else:
                assert len(col) == len(mods) * n_slots
                cols.append(list(col))
        v = [[c[g * n_slots:(g + 1) * n_slots] for c in cols] for g in range(len(mods))]
        return self.biprime_verdict_batch(v, mods)
