#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the UNMODIFIED reference.

Runs only in the build container (needs /root/reference).  The reference cannot be imported
as-is (its un-vendored dependencies tno.mpc.communication / tno.mpc.encryption_schemes.* and
ormsgpack are not installed), so this script pre-seeds ``sys.modules`` with minimal stand-ins
for those *dependencies*  and then imports the reference's own
``paillier_shared_key.py`` and ``distributed_keygen.py`` from where they lie.  The arithmetic
leaf the reference delegates to (pow_mod / mod_inv of tno.mpc.encryption_schemes.utils) is bound
to CPython ``pow`` — which is what that package itself falls back to without gmpy2.

What is recorded (all integers as hex strings):
  * ref_keys.json      — the reference's 24 stored test keys (tests/golden/ref_keys/*.obj, copied
                         data files of the reference's test suite), decoded, plus for every
                         (t, n) key group seeded ciphertexts with the outputs of the reference's
                         PaillierSharedKey.partial_decrypt for every party and of .decrypt.
  * decrypt_synth.json — synthetic keys at key_length 128/1024/2048 (+1 case at 4096) with the
                         reference's partial_decrypt / decrypt outputs, including negative
                         Lagrange exponents and a corrupted-share ValueError case.
  * biprime.json       — outputs of DistributedPaillier.__biprime_test_v_calculation,
                         __biprime_test_with_v_i and __small_prime_divisors_test on seeded
                         candidates (true biprimes and composites) at key_length 64/128/1024/2048.
  * reconstruct.json   — the candidate-modulus step of a keygen round (DK:1262-1284) run by the
                         reference itself: in-process parties over an in-memory pool execute its
                         _generate_pq, `p * q`, `+= zero`, exchange_reconstruct and .reconstruct();
                         recorded are every party's Shamir shares of p, q, zero and of the product
                         (modulo the Shamir prime of DK:647-651) and the reconstructed moduli.

Usage:  python tests/golden/make_golden.py
"""

from __future__ import annotations

import importlib
import json
import math
import os
import random
import sys
import types
from pathlib import Path
from typing import Any, TypedDict

import msgpack
import sympy

HERE = Path(__file__).resolve().parent
REF_PKG = Path("/root/reference/src/tno/mpc/protocols/distributed_keygen")
SEED = 0xD15C0


# --------------------------------------------------------------------------- dependency stand-ins
def _mod(name: str, **attrs: Any) -> types.ModuleType:
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_dependency_standins() -> None:
    for pkg in ("tno", "tno.mpc", "tno.mpc.protocols", "tno.mpc.encryption_schemes"):
        _mod(pkg, __path__=[])
    _mod("tno.mpc.protocols.distributed_keygen", __path__=[str(REF_PKG)])
    _mod(
        "ormsgpack",
        OPT_PASSTHROUGH_BIG_INT=1,
        OPT_PASSTHROUGH_TUPLE=2,
        OPT_PASSTHROUGH_DATACLASS=4,
        OPT_SERIALIZE_NUMPY=8,
        OPT_NON_STR_KEYS=16,
    )

    class Pool:  # noqa: D401 - placeholder
        pass

    class RepetitionError(Exception):
        pass

    class SupportsSerialization:
        pass

    class Serialization:
        @staticmethod
        def register_class(*_a: Any, **_k: Any) -> None:
            return None

    class HTTPClient:
        pass

    _mod(
        "tno.mpc.communication",
        __path__=[],
        Pool=Pool,
        RepetitionError=RepetitionError,
        SupportsSerialization=SupportsSerialization,
        Serialization=Serialization,
    )
    _mod("tno.mpc.communication.pool", Pool=Pool)
    _mod("tno.mpc.communication.httphandlers", HTTPClient=HTTPClient)

    class SecretKey:
        def __init__(self) -> None:
            pass

    class SerializationError(Exception):
        pass

    class EncodedPlaintext:
        def __init__(self, value: Any, scheme: Any = None) -> None:
            self.value = value
            self.scheme = scheme

    _mod(
        "tno.mpc.encryption_schemes.templates",
        __path__=[],
        SecretKey=SecretKey,
        SerializationError=SerializationError,
        EncodedPlaintext=EncodedPlaintext,
    )
    _mod("tno.mpc.encryption_schemes.templates.encryption_scheme", EncodedPlaintext=EncodedPlaintext)
    _mod(
        "tno.mpc.encryption_schemes.utils",
        pow_mod=pow,
        mod_inv=lambda v, m: pow(v, -1, m),
    )

    class _Scheme:
        def __init__(self, number_of_parties: int) -> None:
            self.number_of_parties = number_of_parties

    class IntegerShares:
        def __init__(self, scheme: Any, shares: dict, degree: int, scaling: int) -> None:
            self.scheme = scheme
            self.shares = shares
            self.degree = degree
            self.scaling = scaling

        @property
        def n_fac(self) -> int:
            return math.factorial(self.scheme.number_of_parties)

    class ShamirSecretSharingScheme:
        """Functional stand-in (prime-field Shamir) for the un-vendored scheme: enough for the
        reference's ShamirVariable (utils.py:175-298) to share, add, multiply and reconstruct."""

        def __init__(self, modulus: int, number_of_parties: int, polynomial_degree: int) -> None:
            self.modulus = modulus
            self.number_of_parties = number_of_parties
            self.polynomial_degree = polynomial_degree

        def share_secret(self, secret: int) -> "ShamirShares":
            import secrets as _s

            coeffs = [secret % self.modulus] + [_s.randbelow(self.modulus) for _ in range(self.polynomial_degree)]
            shares = {
                i: sum(c * pow(i, k, self.modulus) for k, c in enumerate(coeffs)) % self.modulus
                for i in range(1, self.number_of_parties + 1)
            }
            return ShamirShares(self, shares)

    class ShamirShares:
        def __init__(self, scheme: Any, shares: dict) -> None:
            self.scheme = scheme
            self.shares = shares

        def __add__(self, other: "ShamirShares") -> "ShamirShares":
            m = self.scheme.modulus
            keys = self.shares.keys() & other.shares.keys()
            deg = max(self.scheme.polynomial_degree, other.scheme.polynomial_degree)
            scheme = ShamirSecretSharingScheme(m, self.scheme.number_of_parties, deg)
            return ShamirShares(scheme, {k: (self.shares[k] + other.shares[k]) % m for k in keys})

        def __mul__(self, other: "ShamirShares") -> "ShamirShares":
            m = self.scheme.modulus
            keys = self.shares.keys() & other.shares.keys()
            scheme = ShamirSecretSharingScheme(
                m, self.scheme.number_of_parties, self.scheme.polynomial_degree + other.scheme.polynomial_degree
            )
            return ShamirShares(scheme, {k: (self.shares[k] * other.shares[k]) % m for k in keys})

        def reconstruct_secret(self) -> int:
            m = self.scheme.modulus
            pts = sorted(self.shares.items())[: self.scheme.polynomial_degree + 1]
            assert len(pts) == self.scheme.polynomial_degree + 1, "not enough shares"
            total = 0
            for i, y in pts:
                num = den = 1
                for j, _ in pts:
                    if j != i:
                        num = num * j % m
                        den = den * (j - i) % m
                total = (total + y * num * pow(den, -1, m)) % m
            return total

    class ShamirSecretSharingIntegers:
        pass

    _mod(
        "tno.mpc.encryption_schemes.shamir",
        IntegerShares=IntegerShares,
        ShamirShares=ShamirShares,
        ShamirSecretSharingScheme=ShamirSecretSharingScheme,
        ShamirSecretSharingIntegers=ShamirSecretSharingIntegers,
        _Scheme=_Scheme,
    )

    class PaillierPublicKey:
        def __init__(self, n: int, g: int) -> None:
            self.n, self.g = n, g

    class PaillierSecretKey:
        pass

    class _PKScheme:
        def __init__(self, n: int) -> None:
            self.public_key = PaillierPublicKey(n, n + 1)

    class PaillierCiphertext:
        def __init__(self, raw_value: int, scheme: Any) -> None:
            self._raw_value = raw_value
            self.scheme = scheme

        def get_value(self) -> int:
            return self._raw_value

        def peek_value(self) -> int:
            return self._raw_value

    class Paillier:
        class SerializedPaillier(TypedDict):
            pass

    inner = _mod(
        "tno.mpc.encryption_schemes.paillier.paillier",
        PaillierCiphertext=PaillierCiphertext,
        Plaintext=Any,
    )
    _mod(
        "tno.mpc.encryption_schemes.paillier",
        __path__=[],
        Paillier=Paillier,
        PaillierCiphertext=PaillierCiphertext,
        PaillierPublicKey=PaillierPublicKey,
        PaillierSecretKey=PaillierSecretKey,
        paillier=inner,
        _PKScheme=_PKScheme,
    )


def load_reference():
    install_dependency_standins()
    psk = importlib.import_module("tno.mpc.protocols.distributed_keygen.paillier_shared_key")
    dk = importlib.import_module("tno.mpc.protocols.distributed_keygen.distributed_keygen")
    return psk, dk


# --------------------------------------------------------------------------- helpers
def hx(v: int) -> str:
    return ("-" if v < 0 else "") + hex(abs(v))


def decode_obj(path: Path) -> dict:
    """Decode one stored key of the reference's test suite (plain msgpack)."""

    def conv(o: Any) -> Any:
        if isinstance(o, dict):
            if set(o.keys()) == {"type", "data"}:
                if o["type"] == "int":
                    return int.from_bytes(o["data"], "little", signed=True)
                return conv(o["data"])
            return {k: conv(v) for k, v in o.items()}
        if isinstance(o, list):
            return [conv(v) for v in o]
        return o

    raw = msgpack.unpackb(path.read_bytes(), strict_map_key=False)
    return conv(raw["object"])


def encrypt(rng: random.Random, m: int, n: int) -> int:
    """Paillier encryption with g = n + 1:  c = (1 + m n) r^n mod n^2."""
    n2 = n * n
    while True:
        r = rng.randrange(1, n)
        if math.gcd(r, n) == 1:
            break
    return (1 + m * n) % n2 * pow(r, n, n2) % n2


def rand_prime(rng: random.Random, bits: int, mod4: int = 3) -> int:
    while True:
        p = sympy.nextprime(rng.getrandbits(bits) | (1 << (bits - 1)))
        if p.bit_length() == bits and p % 4 == mod4:
            return int(p)


def synth_key(rng: random.Random, key_length: int, n_parties: int, t: int, kappa: int = 40) -> dict:
    """A key with the structure DK:1364-1500 produces (integer-Shamir sharing of lambda*beta)."""
    half = key_length // 2
    p, q = rand_prime(rng, half), rand_prime(rng, half)
    while p == q:
        q = rand_prime(rng, half)
    n = p * q
    n_fac = math.factorial(n_parties)
    lam = n - p - q + 1  # DK:1191-1195 summed over parties
    beta = sum(rng.randrange(n) for _ in range(n_parties))  # DK:1449, summed
    bound = (n_fac**2) * (1 << kappa) * n * n_parties  # integer-Shamir coefficient range

    def poly(secret: int) -> list[int]:
        return [n_fac * secret] + [rng.randrange(-bound, bound) for _ in range(t)]

    def ev(coeffs: list[int], x: int) -> int:
        return sum(c * x**k for k, c in enumerate(coeffs))

    fl, fb = poly(lam), poly(beta)
    shares = {i: ev(fl, i) * ev(fb, i) for i in range(1, n_parties + 1)}  # lambda_*beta DK:1465
    theta = (lam * beta * n_fac**3) % n  # DK:1483-1489
    return {
        "key_length": key_length,
        "n_parties": n_parties,
        "t": t,
        "n": n,
        "p": p,
        "q": q,
        "degree": 2 * t,
        "n_fac": n_fac,
        "shares": shares,
        "theta": theta,
    }


def run_decrypt_case(psk, key: dict, ciphertexts: list[int], corrupt: bool = False) -> dict:
    """Push ciphertexts through the reference's PaillierSharedKey objects of every party."""
    shamir = sys.modules["tno.mpc.encryption_schemes.shamir"]
    pail = sys.modules["tno.mpc.encryption_schemes.paillier"]
    n = key["n"]
    parties = sorted(key["shares"].keys())
    keys = {}
    for i in parties:
        share = shamir.IntegerShares(
            shamir._Scheme(key["n_parties"]), {i: key["shares"][i]}, key["degree"], key["n_fac"] ** 2
        )
        keys[i] = psk.PaillierSharedKey(n=n, t=key["t"], player_id=i, share=share, theta=key["theta"])
    scheme = pail._PKScheme(n)
    out = []
    for c in ciphertexts:
        partials = {i: int(keys[i].partial_decrypt(pail.PaillierCiphertext(c, scheme))) for i in parties}
        if corrupt:
            partials[1] = (partials[1] * 3 + 1) % (n * n)
        try:
            m = int(keys[parties[0]].decrypt(dict(partials)))
            err = None
        except ValueError as e:  # PSK:119-123
            m, err = None, "ValueError"
        out.append(
            {
                "c": hx(c),
                "partials": {str(i): hx(v) for i, v in partials.items()},
                "m": None if m is None else hx(m),
                "error": err,
            }
        )
    return {
        "key_length": key.get("key_length"),
        "n_parties": key["n_parties"],
        "t": key["t"],
        "n": hx(n),
        "degree": key["degree"],
        "n_fac": hx(key["n_fac"]),
        "theta": hx(key["theta"]),
        "theta_inv": hx(int(keys[parties[0]].theta_inv)),
        "shares": {str(i): hx(v) for i, v in key["shares"].items()},
        "cases": out,
    }


# --------------------------------------------------------------------------- generators
def gen_ref_keys(psk) -> dict:
    rng = random.Random(SEED + 1)
    groups = {}
    for path in sorted((HERE / "ref_keys").glob("*.obj")):
        obj = decode_obj(path)
        pk = obj["priv_key"]
        t, npar = obj["corruption_threshold"], len(obj["party_indices"])
        g = groups.setdefault(
            (t, npar),
            {
                "n_parties": npar,
                "t": t,
                "n": pk["n"],
                "degree": pk["share"]["degree"],
                "n_fac": math.factorial(pk["share"]["scheme"]["number_of_parties"]),
                "scaling": pk["share"]["scaling"],
                "theta": pk["theta"],
                "shares": {},
                "files": [],
            },
        )
        assert g["n"] == pk["n"] and g["theta"] == pk["theta"]
        (idx, val), = pk["share"]["shares"].items()
        assert idx == pk["player_id"] == obj["index"]
        g["shares"][int(idx)] = val
        g["files"].append(path.name)
    out = {}
    # plaintexts of the reference's own round-trip tests (TDK:20-22), fixed-point encoded with
    # precision 8 as CONF:84 does would need the un-vendored encoder; raw integers are used instead.
    for (t, npar), g in sorted(groups.items()):
        n = g["n"]
        msgs = [0, 1, 2, 3, n - 1, n - 2, n - 3, rng.randrange(n), rng.randrange(n), 42]
        cts = [encrypt(rng, m, n) for m in msgs]
        res = run_decrypt_case(psk, g, cts)
        res["scaling"] = hx(g["scaling"])
        res["files"] = g["files"]
        res["plaintexts"] = [hx(m) for m in msgs]
        for case, m in zip(res["cases"], msgs):
            assert int(case["m"], 16) == m, "reference failed to decrypt its own fixture"
        out[f"t{t}_n{npar}"] = res
    return out


def gen_decrypt_synth(psk) -> dict:
    rng = random.Random(SEED + 2)
    out = {}
    plan = [  # (key_length, n_parties, t, n_ciphertexts)
        (128, 3, 1, 6),
        (128, 5, 2, 4),
        (1024, 3, 1, 4),
        (2048, 3, 1, 4),
        (2048, 5, 2, 2),
        (4096, 3, 1, 1),
    ]
    for kl, npar, t, cnt in plan:
        key = synth_key(rng, kl, npar, t)
        n = key["n"]
        msgs = [0, 1, n - 1] + [rng.randrange(n) for _ in range(max(0, cnt - 3))]
        msgs = msgs[:cnt]
        cts = [encrypt(rng, m, n) for m in msgs]
        res = run_decrypt_case(psk, key, cts)
        res["plaintexts"] = [hx(m) for m in msgs]
        for case, m in zip(res["cases"], msgs):
            assert int(case["m"], 16) == m
        out[f"k{kl}_n{npar}_t{t}"] = res
    # inconsistent partials -> ValueError (PSK:119-123)
    key = synth_key(rng, 128, 3, 1)
    cts = [encrypt(rng, 5, key["n"])]
    res = run_decrypt_case(psk, key, cts, corrupt=True)
    assert res["cases"][0]["error"] == "ValueError"
    out["k128_n3_t1_corrupt"] = res
    return out


def gen_biprime(dk) -> dict:
    rng = random.Random(SEED + 3)
    DP = dk.DistributedPaillier
    v_calc = DP._DistributedPaillier__biprime_test_v_calculation
    verdict = DP._DistributedPaillier__biprime_test_with_v_i
    sieve = DP._DistributedPaillier__small_prime_divisors_test
    utils = sys.modules["tno.mpc.protocols.distributed_keygen.utils"]
    out: dict[str, Any] = {"candidates": [], "sieve": []}

    def additive_split(total: int, npar: int, bits: int, first_mod4: int) -> list[int]:
        """Additive shares of the DK:855-876 shape summing to `total` (p or q)."""
        while True:
            parts = [
                DP._generate_prime_candidate(i + 1, bits) for i in range(npar - 1)
            ]
            last = total - sum(parts)
            if last > 0 and last % 4 == (first_mod4 if npar == 1 else 0):
                return parts + [last]

    def run_candidate(label: str, npar: int, p_parts: list[int], q_parts: list[int], nbip: int, n_g: int) -> None:
        p, q = sum(p_parts), sum(q_parts)
        modulus = p * q
        g_values = [rng.randint(0, modulus) % modulus for _ in range(n_g)]
        party_indices = {f"party{i}": i for i in range(1, npar + 1)}
        per_party = {}
        batched_all = utils.Batched(utils.AdditiveVariable(label="v", modulus=modulus), batch_size=nbip)
        for i in range(1, npar + 1):
            b = v_calc(g_values, i, modulus, p_parts[i - 1], q_parts[i - 1], nbip)
            vals = []
            for var in b.variables:
                try:
                    vals.append(int(var.get_share(i)))
                except KeyError:
                    break
            per_party[i] = vals
            batched_all.set_share(i, vals)
        try:
            ok = bool(verdict(batched_all, modulus, nbip, party_indices))
        except KeyError:
            ok = "KeyError"
        out["candidates"].append(
            {
                "label": label,
                "n_parties": npar,
                "modulus": hx(modulus),
                "p_parts": [hx(x) for x in p_parts],
                "q_parts": [hx(x) for x in q_parts],
                "correct_param_biprime": nbip,
                "g_values": [hx(g) for g in g_values],
                "v": {str(i): [hx(v) for v in vals] for i, vals in per_party.items()},
                "verdict": ok,
            }
        )

    def candidate_parts(npar: int, half_bits: int, want_biprime: bool) -> tuple[list[int], list[int]]:
        # Party shares have `half_bits` bits each (DK:874-876), so p, q have ~half_bits+log2(n) bits.
        while True:
            p_parts = [DP._generate_prime_candidate(i + 1, half_bits) for i in range(npar)]
            q_parts = [DP._generate_prime_candidate(i + 1, half_bits) for i in range(npar)]
            if not want_biprime:
                return p_parts, q_parts
            # steer the last party's share so that p and q are prime (keeps the 0 mod 4 shape)
            ok = True
            for parts in (p_parts, q_parts):
                base = sum(parts)
                tgt = base
                for _ in range(20000):
                    if sympy.isprime(tgt):
                        break
                    tgt += 4
                else:
                    ok = False
                    break
                parts[-1] += tgt - base
                if parts[-1].bit_length() != half_bits:
                    ok = False
                    break
            if ok and sum(p_parts) != sum(q_parts):
                return p_parts, q_parts

    # DK:855-876 uses `secrets`; make it reproducible for this script only.
    import secrets as _secrets

    _secrets.randbits = rng.getrandbits  # type: ignore[assignment]
    dk.secrets.randbits = rng.getrandbits

    plan = [  # (key_length, n_parties, nbip, n_g, n_biprime, n_composite)
        (64, 3, 20, 80, 2, 3),
        (128, 3, 40, 160, 2, 3),
        (128, 5, 40, 160, 1, 1),
        (1024, 3, 40, 160, 1, 2),
        (2048, 3, 40, 160, 1, 1),
        (2048, 5, 40, 160, 0, 1),
    ]
    for kl, npar, nbip, n_g, nb, nc in plan:
        for k in range(nb):
            pp, qp = candidate_parts(npar, kl // 2, True)
            run_candidate(f"k{kl}_n{npar}_biprime{k}", npar, pp, qp, nbip, n_g)
        for k in range(nc):
            pp, qp = candidate_parts(npar, kl // 2, False)
            run_candidate(f"k{kl}_n{npar}_composite{k}", npar, pp, qp, nbip, n_g)
    # a case with too few Jacobi-1 generators: reference raises KeyError in the verdict
    pp, qp = candidate_parts(3, 32, True)
    run_candidate("k64_n3_short_g", 3, pp, qp, 20, 12)

    # sieve: DK:552-554 prime lists and DK:1197-1209 verdicts
    for threshold in (200, 2000, 20000):
        primes = [int(x) for x in sympy.primerange(3, threshold + 1)]
        cases = []
        for bits in (68, 131, 1027, 2051):
            for _ in range(6):
                cand = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
                cases.append({"modulus": hx(cand), "has_small_divisor": bool(sieve(primes, cand))})
            # candidates with no small divisor / with exactly the largest prime as divisor
            big = int(sympy.nextprime(1 << (bits // 2))) * int(sympy.nextprime((1 << (bits - bits // 2)) + 12345))
            cases.append({"modulus": hx(big), "has_small_divisor": bool(sieve(primes, big))})
            last = primes[-1] * int(sympy.nextprime(1 << (bits - 16)))
            cases.append({"modulus": hx(last), "has_small_divisor": bool(sieve(primes, last))})
        out["sieve"].append({"prime_threshold": threshold, "n_primes": len(primes),
                             "first": primes[:3], "last": primes[-1], "cases": cases})
    return out


# --------------------------------------------------------------------------- N reconstruction (DK:1262-1284)
class _Hub:
    def __init__(self, names):
        self.names = names
        self.box = {n: {} for n in names}


class _MemPool:
    """The calls of tno.mpc.communication.Pool that _generate_pq / exchange_* use, in memory."""

    def __init__(self, hub, me):
        self.hub, self.me = hub, me
        self.pool_handlers = {n: None for n in hub.names if n != me}

    def asend(self, party, message, msg_id=None):
        self.hub.box[party].setdefault(msg_id, []).append((self.me, message))

    def async_broadcast(self, message, msg_id=None, handler_names=None):
        for n in (handler_names if handler_names is not None else self.pool_handlers):
            self.hub.box[n].setdefault(msg_id, []).append((self.me, message))

    async def recv_all(self, msg_id=None):
        import asyncio

        while len(self.hub.box[self.me].get(msg_id, [])) < len(self.pool_handlers):
            await asyncio.sleep(0)
        return tuple(self.hub.box[self.me].pop(msg_id))


def gen_reconstruct(dk) -> dict:
    import asyncio
    import secrets as _secrets

    DP = dk.DistributedPaillier
    out = {}
    for label, key_length, n_parties, t, batch in (
        ("k64_n3_t1", 64, 3, 1, 6), ("k128_n5_t2", 128, 5, 2, 4), ("k1024_n3_t1", 1024, 3, 1, 3), ("k2048_n5_t2", 2048, 5, 2, 2),
    ):
        rng = random.Random(SEED + key_length * 7 + n_parties)
        saved = (_secrets.randbits, _secrets.randbelow)
        _secrets.randbits = rng.getrandbits
        _secrets.randbelow = lambda n: rng.randrange(n)
        dk.secrets.randbits = rng.getrandbits
        try:
            names = [f"p{i}" for i in range(1, n_parties + 1)]
            hub = _Hub(names)
            record = {}

            async def party(i, me):
                pool = _MemPool(hub, me)
                party_indices = {("self" if n == me else n): k for k, n in enumerate(names, start=1)}
                _, prime_length, _, sh_t, sh_2t, _ = DP.setup_input(pool, key_length, 200, t)
                p_sh, q_sh, zero, p_add, q_add = await DP._generate_pq(
                    pool, i, prime_length, party_indices, sh_t, sh_2t, 99, batch_size=batch, msg_id=f"pq_{label}")
                candidate_n = p_sh * q_sh                     # DK:1274
                candidate_n += zero                           # DK:1277
                mine = {
                    "p": [v.get_share(i) for v in p_sh.variables], "q": [v.get_share(i) for v in q_sh.variables],
                    "zero": [v.get_share(i) for v in zero.variables], "n": [v.get_share(i) for v in candidate_n.variables],
                }
                await dk.exchange_reconstruct(candidate_n, i, pool, party_indices, msg_id=f"n_{label}")   # DK:1281
                moduli = candidate_n.reconstruct()            # DK:1284
                record[i] = (mine, moduli, sh_t.modulus, [dict(v.get_shares()) for v in candidate_n.variables],
                             [int(x) for x in p_add], [int(x) for x in q_add])

            async def run():
                await asyncio.gather(*[party(i, me) for i, me in enumerate(names, start=1)])

            asyncio.run(run())
        finally:
            _secrets.randbits, _secrets.randbelow = saved
            dk.secrets.randbits = saved[0]
        moduli = record[1][1]
        assert all(record[i][1] == moduli for i in record), "parties reconstructed different moduli"
        prime = record[1][2]
        assert int(prime) == int(sympy.nextprime(2 ** (2 * (key_length // 2 + math.ceil(math.log2(n_parties))))))   # DK:647-651
        for i in record:                                       # every party saw the same share table
            assert record[i][3] == record[1][3]
        # The pin that does not depend on the Shamir stand-in: the additive shares p_i, q_i come from the reference's own
        # candidate generation (DK:854-876, its secrets-based sampling), and whatever Shamir implementation carries them,
        # the modulus of candidate k is (sum_i p_i[k]) * (sum_i q_i[k]).
        for k, m in enumerate(moduli):
            assert m == sum(record[i][4][k] for i in record) * sum(record[i][5][k] for i in record), (label, k)
        out[label] = {
            "key_length": key_length, "n_parties": n_parties, "t": t, "degree": 2 * t, "prime": hx(int(prime)),
            "shares": {str(i): {k: [hx(v) for v in vals] for k, vals in record[i][0].items()} for i in sorted(record)},
            "moduli": [hx(m) for m in moduli],
            "p_additive": {str(i): [hx(v) for v in record[i][4]] for i in sorted(record)},
            "q_additive": {str(i): [hx(v) for v in record[i][5]] for i in sorted(record)},
            "provenance": {
                "reference_code": "DistributedPaillier.setup_input (Shamir prime, DK:647-651), _generate_pq (additive shares "
                                  "p_additive / q_additive, DK:854-876; the share-and-exchange flow DK:720-852), Batched / "
                                  "ShamirVariable containers (utils.py:175-298, 386-500: which share is multiplied with "
                                  "which, `+= zero`), exchange_reconstruct (utils.py:560-594)",
                "stand_in_code": "tno.mpc.encryption_schemes.shamir is not vendored: ShamirSecretSharingScheme.share_secret "
                                 "(random polynomial, evaluation at 1..n), ShamirShares.__add__ / __mul__ (share-wise modulo "
                                 "the prime) and .reconstruct_secret (Lagrange at 0 over the first degree+1 points) are the "
                                 "textbook stand-in of this script — so `shares` (p, q, zero, n) are stand-in numbers",
                "pinned_by_reference_alone": "`moduli` == (sum p_additive) * (sum q_additive) per candidate (asserted when "
                                             "generated): both factors come from the reference's own sampling",
            },
        }
    return out


def main() -> None:
    psk, dk = load_reference()
    os.makedirs(HERE, exist_ok=True)
    for name, data in (
        ("ref_keys.json", gen_ref_keys(psk)),
        ("decrypt_synth.json", gen_decrypt_synth(psk)),
        ("biprime.json", gen_biprime(dk)),
        ("reconstruct.json", gen_reconstruct(dk)),
    ):
        (HERE / name).write_text(json.dumps(data, indent=0, sort_keys=True) + "\n")
        print("wrote", name, (HERE / name).stat().st_size, "bytes")


if __name__ == "__main__":
    main()
