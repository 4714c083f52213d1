"""Installs the GPU engine into an importable reference package (drop-in patch).

``install()`` rebinds, by ``setattr`` on the reference's own classes (the technique its benchmark
script uses at scripts/bench_batch_size.py:94-110):

  PaillierSharedKey.partial_decrypt / .decrypt           paillier_shared_key.py:52-127
      -> GPU-backed, plus new .partial_decrypt_batch / .decrypt_batch
  DistributedPaillier._decrypt_sequence_raw              distributed_keygen.py:430-517
      -> same message flow (the reference's own coroutine still runs), but its two loops
         (:463-466 partial decryptions, :510-515 recombinations) execute as ONE launch each
  DistributedPaillier.__small_prime_divisors_test        :1197-1209
  DistributedPaillier.__biprime_test_v_calculation       :1056-1108
  DistributedPaillier.__biprime_test_with_v_i            :1110-1175
      -> GPU-backed scalar forms (same signatures and return types)
  DistributedPaillier.compute_modulus                    :1211-1362
      -> same rounds and messages, with the sieve, the v-calculation and the verdict of a round
         each executed as ONE launch over the round's candidates

so ``DistributedPaillier.from_security_parameter()``, ``.decrypt()`` and ``.decrypt_sequence()``
work unmodified.  ``uninstall()`` restores the originals.  The control plane (pools, message ids,
share containers) is the reference's and is not re-implemented here.
"""

from __future__ import annotations

import importlib
from typing import Any, Dict, Iterable, List, Optional

from . import biprime
from .shared_key import GpuPaillierSharedKey

DEFAULT_PACKAGE = "tno.mpc.protocols.distributed_keygen"
_saved: Dict[Any, Dict[str, Any]] = {}


def _save(cls: Any, name: str) -> None:
    _saved.setdefault(cls, {})
    if name not in _saved[cls]:
        _saved[cls][name] = cls.__dict__.get(name, None)


def _gpu_key(key: Any, engine: Any) -> GpuPaillierSharedKey:
    cached = getattr(key, "_mx_gpu_key", None)
    if cached is None or cached.n != key.n or cached.share is not key.share:
        cached = GpuPaillierSharedKey.from_reference(key, engine)
        key._mx_gpu_key = cached
    return cached


class _Deferred(int):
    """Placeholder returned by PaillierSharedKey.decrypt while a sequence is being recombined."""


def install(engine: Any = None, package: str = DEFAULT_PACKAGE) -> None:
    psk_mod = importlib.import_module(package + ".paillier_shared_key")
    dk_mod = importlib.import_module(package + ".distributed_keygen")
    PSK = psk_mod.PaillierSharedKey
    DP = dk_mod.DistributedPaillier

    # ------------------------------------------------------------------ PaillierSharedKey
    def partial_decrypt_batch(self: Any, ciphertexts: Iterable[Any]) -> List[int]:
        return _gpu_key(self, engine).partial_decrypt_batch(ciphertexts)

    def decrypt_batch(self: Any, partial_dicts: List[Dict[int, int]]) -> List[int]:
        return _gpu_key(self, engine).decrypt_batch(partial_dicts)

    def partial_decrypt(self: Any, ciphertext: Any) -> int:
        cache = getattr(self, "_mx_partial_cache", None)
        if cache is not None and id(ciphertext) in cache:
            return cache.pop(id(ciphertext))
        return partial_decrypt_batch(self, [ciphertext])[0]

    def decrypt(self: Any, partial_dict: Dict[int, int]) -> int:
        pending = getattr(self, "_mx_pending_combines", None)
        if pending is not None:
            pending.append(partial_dict)
            return _Deferred(len(pending) - 1)
        return decrypt_batch(self, [partial_dict])[0]

    for name, fn in (
        ("partial_decrypt", partial_decrypt),
        ("decrypt", decrypt),
        ("partial_decrypt_batch", partial_decrypt_batch),
        ("decrypt_batch", decrypt_batch),
    ):
        _save(PSK, name)
        setattr(PSK, name, fn)

    # ------------------------------------------------------------------ decrypt_sequence
    _save(DP, "_decrypt_sequence_raw")
    original_sequence = _saved[DP]["_decrypt_sequence_raw"]

    async def _decrypt_sequence_raw(self: Any, ciphertext_sequence: Iterable[Any], receivers: Optional[List[str]] = None):
        sequence = list(ciphertext_sequence)
        key = self.secret_key
        # loop DK:463-466 as one launch; the reference's coroutine then finds every result cached
        partials = key.partial_decrypt_batch(sequence)
        key._mx_partial_cache = {id(c): p for c, p in zip(sequence, partials)}
        key._mx_pending_combines = []
        try:
            result = await original_sequence(self, sequence, receivers)
            pending = key._mx_pending_combines
        finally:
            key._mx_partial_cache = None
            key._mx_pending_combines = None
        if result is not None and pending:
            # loop DK:510-515 as one launch; fill the placeholders the coroutine wrapped
            messages = key.decrypt_batch(pending)
            for encoded in result:
                if isinstance(encoded.value, _Deferred):
                    encoded.value = messages[int(encoded.value)]
        return result

    setattr(DP, "_decrypt_sequence_raw", _decrypt_sequence_raw)

    # ------------------------------------------------------------------ keygen class-methods
    Batched, AdditiveVariable = dk_mod.Batched, dk_mod.AdditiveVariable

    def _to_batched(values: List[int], index: int, modulus: int, slots: int) -> Any:
        batched = Batched(AdditiveVariable(label="v", modulus=modulus), batch_size=slots)  # DK:1103-1107
        batched.set_share(index, values)
        return batched

    def _v_lists(batched_v_i: Any, party_indices: Dict[str, int]) -> Dict[int, List[int]]:
        out: Dict[int, List[int]] = {i: [] for i in party_indices.values()}
        for i in out:
            for var in batched_v_i.variables:
                try:
                    out[i].append(var.get_share(i))
                except KeyError:
                    break
        return out

    def small_prime_divisors_test(cls: Any, prime_list: List[int], modulus: int) -> bool:
        return biprime.small_prime_divisors_test(prime_list, modulus, engine)

    def biprime_test_v_calculation(cls: Any, g_values, index, modulus, p_i, q_i, correct_param_biprime) -> Any:
        values = biprime.biprime_test_v_calculation(g_values, index, modulus, p_i, q_i, correct_param_biprime, engine)
        return _to_batched(values, index, modulus, correct_param_biprime)

    def biprime_test_with_v_i(cls: Any, batched_v_i, modulus, correct_param_biprime, party_indices) -> bool:
        return biprime.biprime_test_with_v_i(_v_lists(batched_v_i, party_indices), modulus, correct_param_biprime, engine)

    mangled = "_DistributedPaillier__"
    for name, fn in (
        (mangled + "small_prime_divisors_test", small_prime_divisors_test),
        (mangled + "biprime_test_v_calculation", biprime_test_v_calculation),
        (mangled + "biprime_test_with_v_i", biprime_test_with_v_i),
    ):
        _save(DP, name)
        setattr(DP, name, classmethod(fn))

    # ------------------------------------------------------------------ compute_modulus, batched per round
    _save(DP, "compute_modulus")
    exchange_reconstruct = dk_mod.exchange_reconstruct
    Shares = dk_mod.Shares
    logger = dk_mod.logger

    async def compute_modulus(
        cls: Any, shares, index, pool, prime_list, party_indices, prime_length, shamir_scheme_t,
        shamir_scheme_2t, correct_param_biprime, session_id, batch_size: int = 1,
    ) -> int:
        sieved_out = biprime_rejected = rounds = 0
        sid = f"distributed_keygen_session#{session_id}"
        while True:
            rounds += 1
            p_sh, q_sh, zero, p_add, q_add = await cls._generate_pq(
                pool, index, prime_length, party_indices, shamir_scheme_t, shamir_scheme_2t, session_id,
                batch_size=batch_size, msg_id=f"{sid}_generate_pq_{rounds}",
            )
            candidate_n = p_sh * q_sh
            candidate_n += zero
            await exchange_reconstruct(candidate_n, index, pool, party_indices, msg_id=f"{sid}_n_{rounds}")
            moduli = candidate_n.reconstruct()
            # DK:1288-1292 as one launch
            has_divisor = biprime.small_prime_divisors_test_batch(prime_list, moduli, engine)
            survivors = [k for k, bad in enumerate(has_divisor) if not bad]
            sieved_out += len(moduli) - len(survivors)
            if not survivors:
                continue
            g_values = await getattr(cls, mangled + "biprime_test_g_generation")(
                correct_param_biprime, index, [moduli[k] for k in survivors], party_indices, pool,
                f"{sid}_biprime_test_g_{rounds}",
            )
            # DK:1313-1329 as one launch
            v_lists = biprime.biprime_test_v_calculation_batch(
                g_values, index, [moduli[k] for k in survivors], [p_add[k] for k in survivors],
                [q_add[k] for k in survivors], correct_param_biprime, engine,
            )
            to_exchange = [
                _to_batched(v, index, moduli[k], correct_param_biprime) for v, k in zip(v_lists, survivors)
            ]
            await exchange_reconstruct(to_exchange, index, pool, party_indices, msg_id=f"{sid}_biprime_test_v_{rounds}_v")
            # DK:1339-1360: slot tests of all survivors as one launch; first passing candidate wins
            verdicts = biprime.biprime_test_with_v_i_batch(
                [_v_lists(b, party_indices) for b in to_exchange], [moduli[k] for k in survivors],
                correct_param_biprime, engine, errors="return",
            )
            for verdict, k in zip(verdicts, survivors):
                shares.p = Shares.P(p_add[k], q_sh[k].get_shares())  # as DK:1344-1345 (sic)
                shares.q = Shares.Q(q_add[k], q_sh[k].get_shares())
                if isinstance(verdict, Exception):
                    raise verdict
                if verdict:
                    logger.info(f"N = {moduli[k]}")
                    logger.info(f"Checked {sieved_out} primes for small prime divisors in {rounds} rounds")
                    logger.info(f"Checked {biprime_rejected} candidates for biprimality")
                    return moduli[k]
                biprime_rejected += 1

    setattr(DP, "compute_modulus", classmethod(compute_modulus))


def uninstall() -> None:
    for cls, names in _saved.items():
        for name, orig in names.items():
            if orig is None:
                if name in cls.__dict__:
                    delattr(cls, name)
            else:
                setattr(cls, name, orig)
    _saved.clear()
