"""Stand-in for the reference's distributed_keygen module (see the package docstring): the class surface
patch.py rebinds, with a minimal three-phase protocol behind it.

    DistributedPaillier._decrypt_sequence_raw / _decrypt_raw      partial decryptions, broadcast, recombination
    DistributedPaillier.compute_modulus                           rounds of: candidate shares -> N -> sieve ->
                                                                  jointly random generators -> v values -> verdict
    _DistributedPaillier__small_prime_divisors_test / __biprime_test_v_calculation / __biprime_test_with_v_i /
    __biprime_test_g_generation, _generate_pq, Batched, AdditiveVariable, ShamirVariable, exchange_reconstruct,
    Shares, EncodedPlaintext, pow_mod, mod_inv, logger

Everything is textbook arithmetic on CPython integers (sympy only for the Jacobi symbol).  Randomness comes
from the module-level ``rng`` so that a patched and an unpatched run can be given the same candidates.
"""

from __future__ import annotations

import asyncio
import hashlib
import logging
import random
from dataclasses import dataclass
from typing import Any, Dict, Iterable, List, Optional, Sequence

from .paillier_shared_key import PaillierSharedKey, mod_inv, pow_mod  # noqa: F401  (leaf names, rebound by install(leaf=True))

logger = logging.getLogger("keygen_standin")
rng = random.Random(0)


class EncodedPlaintext:
    def __init__(self, value: int, scheme: Any = None) -> None:
        self.value, self.scheme = value, scheme


# ---------------------------------------------------------------------------------------------- share containers
class AdditiveVariable:
    def __init__(self, label: str, modulus: int) -> None:
        self.label, self.modulus, self._sharing = label, modulus, {}

    def set_share(self, index: int, value: int) -> None:
        self._sharing[index] = value

    def get_share(self, index: int) -> int:
        return self._sharing[index]            # KeyError on an unset slot

    def clone(self) -> "AdditiveVariable":
        return AdditiveVariable(self.label, self.modulus)


@dataclass(frozen=True)
class ShamirScheme:
    modulus: int
    number_of_parties: int
    polynomial_degree: int

    def share_secret(self, secret: int) -> Dict[int, int]:
        coeffs = [secret % self.modulus] + [rng.randrange(self.modulus) for _ in range(self.polynomial_degree)]
        return {x: sum(c * pow(x, k, self.modulus) for k, c in enumerate(coeffs)) % self.modulus
                for x in range(1, self.number_of_parties + 1)}


class ShamirVariable:
    """One party's view of a Shamir-shared value: the shares it knows, by party index."""

    def __init__(self, shamir_scheme: ShamirScheme, label: str = "") -> None:
        self.shamir_scheme, self.label, self._sharing = shamir_scheme, label, {}

    def set_share(self, index: int, value: int) -> None:
        self._sharing[index] = value % self.shamir_scheme.modulus

    def get_share(self, index: int) -> int:
        return self._sharing[index]

    def get_shares(self) -> Dict[int, int]:
        return self._sharing

    def clone(self) -> "ShamirVariable":
        return ShamirVariable(self.shamir_scheme, self.label)

    def __mul__(self, other: "ShamirVariable") -> "ShamirVariable":
        a, b = self.shamir_scheme, other.shamir_scheme
        out = ShamirVariable(ShamirScheme(a.modulus, a.number_of_parties, a.polynomial_degree + b.polynomial_degree), self.label)
        for i, v in self._sharing.items():
            if i in other._sharing:
                out._sharing[i] = v * other._sharing[i] % a.modulus
        return out

    def __iadd__(self, other: "ShamirVariable") -> "ShamirVariable":
        for i in list(self._sharing):
            self._sharing[i] = (self._sharing[i] + other._sharing[i]) % self.shamir_scheme.modulus
        return self

    def reconstruct(self) -> int:
        sch = self.shamir_scheme
        pts = list(self._sharing)[: sch.polynomial_degree + 1]
        if len(pts) < sch.polynomial_degree + 1:
            raise ValueError("not enough shares")
        total = 0
        for i in pts:
            num = den = 1
            for j in pts:
                if j != i:
                    num = num * j % sch.modulus
                    den = den * (j - i) % sch.modulus
            total += self._sharing[i] * num * pow(den, -1, sch.modulus)
        return total % sch.modulus


class Batched:
    """`batch_size` variables of one kind, handled together."""

    def __init__(self, variable: Any, batch_size: int) -> None:
        self.variables = [variable] + [variable.clone() for _ in range(batch_size - 1)]
        self.batch_size = batch_size

    def set_share(self, index: int, values: Sequence[int]) -> None:
        for var, v in zip(self.variables, values):        # fewer values than slots: the rest stay unset
            var.set_share(index, v)

    def get_share(self, index: int) -> List[int]:
        return [v.get_share(index) for v in self.variables]

    def __getitem__(self, k: int) -> Any:
        return self.variables[k]

    def __mul__(self, other: "Batched") -> "Batched":
        out = Batched.__new__(Batched)
        out.variables = [a * b for a, b in zip(self.variables, other.variables)]
        out.batch_size = self.batch_size
        return out

    def __iadd__(self, other: "Batched") -> "Batched":
        for a, b in zip(self.variables, other.variables):
            a += b
        return self

    def reconstruct(self) -> List[int]:
        return [v.reconstruct() for v in self.variables]


class Shares:
    @dataclass
    class P:
        additive: int
        shares: Dict[int, int]

    @dataclass
    class Q:
        additive: int
        shares: Dict[int, int]

    def __init__(self) -> None:
        self.p: Optional[Shares.P] = None
        self.q: Optional[Shares.Q] = None


async def exchange_reconstruct(variables: Any, index: int, pool: Any, party_indices: Dict[str, int], msg_id: str) -> None:
    """Every party sends its own share of every variable to all others and stores what it receives."""
    batches = variables if isinstance(variables, list) else [variables]
    own = [[v._sharing.get(index) for v in b.variables] for b in batches]
    pool.async_broadcast({"content": "shares", "value": own}, msg_id=msg_id)
    for party, message in await pool.recv_all(msg_id=msg_id):
        j = party_indices[party]
        for b, vals in zip(batches, message["value"]):
            for var, v in zip(b.variables, vals):
                if v is not None:
                    var.set_share(j, v)


# ---------------------------------------------------------------------------------------------- the scheme
class DistributedPaillier:
    default_prime_threshold = 2000
    default_biprime_param = 40

    # ---- decryption
    async def _decrypt_raw(self, ciphertext: Any, receivers: Optional[List[str]] = None):
        """One ciphertext: a message of its own kind ("partial_decryption", a bare value) under an id made of the
        ciphertext's leading bits only — the shape of the reference's single-ciphertext protocol."""
        self_receive = receivers is None or "self" in receivers
        others = None if receivers is None else [r for r in receivers if r != "self"]
        mine = self.secret_key.partial_decrypt(ciphertext)
        msg_id = f"distributed_decryption_session#{self.session_id}_hash#{bin(ciphertext.peek_value()).zfill(32)[2:34]}"
        if others is None or others:
            self.pool.async_broadcast({"content": "partial_decryption", "value": mine}, msg_id=msg_id, handler_names=others)
        if not self_receive:
            return None
        collected = {self.index: mine}
        for party, message in await self.pool.recv_all(msg_id=msg_id):
            assert message["content"] == "partial_decryption"
            collected[self.party_indices[party]] = message["value"]
        return EncodedPlaintext(self.secret_key.decrypt(collected), scheme=self)

    async def _decrypt_sequence_raw(self, ciphertext_sequence: Iterable[Any], receivers: Optional[List[str]] = None):
        sequence = list(ciphertext_sequence)
        self_receive = receivers is None or "self" in receivers
        others = None if receivers is None else [r for r in receivers if r != "self"]
        partials = [self.secret_key.partial_decrypt(c) for c in sequence]
        tag = bin(sequence[0].peek_value()).zfill(32)[2:34] + f"{len(partials)}"
        msg_id = f"distributed_decryption_session#{self.session_id}_hash#{tag}"
        if others is None or others:
            self.pool.async_broadcast({"content": "partial_decryption_sequence", "value": partials}, msg_id=msg_id, handler_names=others)
        if not self_receive:
            return None
        dicts = [{self.index: p} for p in partials]
        for party, message in await self.pool.recv_all(msg_id=msg_id):
            assert message["content"] == "partial_decryption_sequence"
            for d, v in zip(dicts, message["value"]):
                d[self.party_indices[party]] = v
        return [EncodedPlaintext(self.secret_key.decrypt(d), scheme=self) for d in dicts]

    # ---- key generation: inputs
    @classmethod
    def setup_input(cls, n_parties: int, key_length: int, prime_threshold: int, t: int):
        import sympy

        prime_length = key_length // 2
        shamir_length = 2 * (prime_length + (n_parties - 1).bit_length() + 1) + 40
        prime = int(sympy.nextprime(1 << shamir_length))
        prime_list = [int(p) for p in sympy.primerange(3, prime_threshold + 1)]
        return prime_length, prime_list, ShamirScheme(prime, n_parties, t), ShamirScheme(prime, n_parties, 2 * t), Shares()

    @classmethod
    async def _generate_pq(cls, pool, index, prime_length, party_indices, shamir_scheme_t, shamir_scheme_2t, session_id,
                           batch_size: int = 1, msg_id: str = ""):
        """Additive shares of the two prime candidates (party 1's are 3 mod 4, the others' 0 mod 4), their
        Shamir sharings of degree t, and a degree-2t sharing of zero."""
        def additive() -> int:
            v = rng.getrandbits(prime_length - 1) | (1 << (prime_length - 2))
            return (v & ~3) | (3 if index == 1 else 0)

        p_add = [additive() for _ in range(batch_size)]
        q_add = [additive() for _ in range(batch_size)]
        mine = {"p": [shamir_scheme_t.share_secret(v) for v in p_add], "q": [shamir_scheme_t.share_secret(v) for v in q_add],
                "z": [shamir_scheme_2t.share_secret(0) for _ in range(batch_size)]}
        for name, j in party_indices.items():
            if name != "self":
                pool.asend(name, {"content": "pq", "value": {k: [s[j] for s in v] for k, v in mine.items()}}, msg_id=msg_id)
        acc = {k: [s[index] for s in v] for k, v in mine.items()}
        for _, message in await pool.recv_all(msg_id=msg_id):
            for k in acc:
                acc[k] = [a + b for a, b in zip(acc[k], message["value"][k])]
        out = []
        for k, scheme in (("p", shamir_scheme_t), ("q", shamir_scheme_t), ("z", shamir_scheme_2t)):
            b = Batched(ShamirVariable(scheme, k), batch_size)
            b.set_share(index, acc[k])
            out.append(b)
        return out[0], out[1], out[2], p_add, q_add

    @classmethod
    async def __biprime_test_g_generation(cls, correct_param_biprime, index, candidate_n_list, party_indices, pool, msg_id):
        """4 x correct_param jointly random values in [0, N) per modulus: every party contributes a seed, the
        values are derived from all seeds (identical on every party)."""
        seed = rng.getrandbits(128)
        pool.async_broadcast({"content": "g_seed", "value": seed}, msg_id=msg_id)
        seeds = {index: seed}
        for party, message in await pool.recv_all(msg_id=msg_id):
            seeds[party_indices[party]] = message["value"]
        joint = hashlib.sha256(repr(sorted(seeds.items())).encode()).digest()
        g_rng = random.Random(joint)
        return [[g_rng.randrange(n) for _ in range(4 * correct_param_biprime)] for n in candidate_n_list]

    # ---- key generation: the three tests
    @classmethod
    def __small_prime_divisors_test(cls, prime_list: List[int], modulus: int) -> bool:
        return any(modulus % p == 0 for p in prime_list)

    @classmethod
    def __biprime_test_v_calculation(cls, g_values, index, modulus, p_i, q_i, correct_param_biprime):
        import sympy

        exponent = (modulus - p_i - q_i + 1) // 4 if index == 1 else (p_i + q_i) // 4
        values: List[int] = []
        for g in g_values:
            if len(values) == correct_param_biprime:
                break
            if sympy.jacobi_symbol(g, modulus) == 1:
                values.append(int(pow_mod(g, exponent, modulus)))
        batched = Batched(AdditiveVariable(label="v", modulus=modulus), batch_size=correct_param_biprime)
        batched.set_share(index, values)
        return batched

    @classmethod
    def __biprime_test_with_v_i(cls, batched_v_i, modulus, correct_param_biprime, party_indices) -> bool:
        passed = 0
        for var in batched_v_i.variables:
            product = 1
            for i in party_indices.values():
                if i != 1:
                    product *= var.get_share(i)
            v1 = var.get_share(1) % modulus
            if v1 != product % modulus and v1 != (-product) % modulus:
                return False
            passed += 1
            if passed >= correct_param_biprime:
                return True
        return False

    # ---- key generation: the loop
    @classmethod
    async def compute_modulus(cls, shares, index, pool, prime_list, party_indices, prime_length, shamir_scheme_t,
                              shamir_scheme_2t, correct_param_biprime, session_id, batch_size: int = 1) -> int:
        rounds = 0
        sid = f"distributed_keygen_session#{session_id}"
        while True:
            rounds += 1
            p_sh, q_sh, zero, p_add, q_add = await cls._generate_pq(
                pool, index, prime_length, party_indices, shamir_scheme_t, shamir_scheme_2t, session_id,
                batch_size=batch_size, msg_id=f"{sid}_generate_pq_{rounds}")
            candidate_n = p_sh * q_sh
            candidate_n += zero
            await exchange_reconstruct(candidate_n, index, pool, party_indices, msg_id=f"{sid}_n_{rounds}")
            moduli = candidate_n.reconstruct()
            survivors = [k for k, n in enumerate(moduli) if not cls.__small_prime_divisors_test(prime_list, n)]
            if not survivors:
                continue
            g_values = await cls.__biprime_test_g_generation(
                correct_param_biprime, index, [moduli[k] for k in survivors], party_indices, pool, f"{sid}_biprime_test_g_{rounds}")
            to_exchange = [cls.__biprime_test_v_calculation(g, index, moduli[k], p_add[k], q_add[k], correct_param_biprime)
                           for g, k in zip(g_values, survivors)]
            await exchange_reconstruct(to_exchange, index, pool, party_indices, msg_id=f"{sid}_biprime_test_v_{rounds}_v")
            for batched, k in zip(to_exchange, survivors):
                shares.p = Shares.P(p_add[k], q_sh[k].get_shares())
                shares.q = Shares.Q(q_add[k], q_sh[k].get_shares())
                if cls.__biprime_test_with_v_i(batched, moduli[k], correct_param_biprime, party_indices):
                    logger.info(f"N = {moduli[k]} after {rounds} rounds")
                    return moduli[k]


# ---------------------------------------------------------------------------------------------- in-memory pool
class Hub:
    def __init__(self, names: Sequence[str], wire: Optional[Any] = None) -> None:
        self.names, self.box, self.wire = list(names), {n: {} for n in names}, wire

    def pool(self, me: str) -> "MemoryPool":
        return MemoryPool(self, me)


class MemoryPool:
    """The calls of the un-vendored communication pool that the two protocols use.  With ``hub.wire`` set,
    partial-decryption lists travel through it (e.g. codec.encode_int per element: what arrives then is the
    serialised form ``{"type": "int", "data": bytes}`` a real transport delivers)."""

    def __init__(self, hub: Hub, me: str) -> None:
        self.hub, self.me = hub, me
        self.pool_handlers = {n: None for n in hub.names if n != me}

    def _deliver(self, to: str, message: Dict[str, Any], msg_id: str) -> None:
        if self.hub.wire is not None and message.get("content") == "partial_decryption_sequence":
            message = dict(message, value=self.hub.wire(self.me, message["value"]))
        self.hub.box[to].setdefault(msg_id, []).append((self.me, message))

    def async_broadcast(self, message, msg_id=None, handler_names=None) -> None:
        for n in (handler_names if handler_names is not None else self.pool_handlers):
            self._deliver(n, message, msg_id)

    def asend(self, party, message, msg_id=None) -> None:
        self._deliver(party, message, msg_id)

    async def recv_all(self, msg_id=None):
        while len(self.hub.box[self.me].get(msg_id, [])) < len(self.pool_handlers):
            await asyncio.sleep(0)
        return tuple(self.hub.box[self.me].pop(msg_id))
