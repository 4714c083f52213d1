"""Installs the GPU engine into an importable reference package (drop-in patch).

``install()`` rebinds, by ``setattr`` on the reference's own classes (the technique its benchmark
script uses at scripts/bench_batch_size.py:94-110):

  PaillierSharedKey.partial_decrypt / .decrypt           paillier_shared_key.py:52-127
      -> GPU-backed, plus new .partial_decrypt_batch / .decrypt_batch
  DistributedPaillier._decrypt_raw                       distributed_keygen.py:314-382
      -> same receivers logic, message id and message content; its two arithmetic call sites (:345-349 partial
         decryption, :378-380 recombination) are awaits on a per-process micro-batcher (coalesce.Coalescer): every
         ``decrypt()`` coroutine pending in the same turn of the event loop shares ONE launch per key
  DistributedPaillier._decrypt_sequence_raw              distributed_keygen.py:430-517
      -> same receivers logic, message id and message content, with its two loops (:463-466 partial
         decryptions, :510-515 recombinations) executed as ONE launch each; received partial
         decryptions go to the device column-wise (codec.rows_from_wire), the party's own stay there
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

from . import biprime, shamir
from .coalesce import Coalescer, RoundCoalescer
from .shared_key import GpuPaillierSharedKey

DEFAULT_PACKAGE = "tno.mpc.protocols.distributed_keygen"
_saved: Dict[Any, Dict[str, Any]] = {}
_saved_names: Dict[Any, Dict[str, Any]] = {}      # module -> {name: original leaf function}
_warned_late_queues = False                        # the "too late for hardware queues" warning is given once per process
_coalescers: Dict[str, Coalescer] = {}            # package -> the micro-batcher of its installed patch
_round_coalescers: Dict[str, RoundCoalescer] = {}  # package -> the batcher of the keygen rounds of co-located parties


def round_coalescer(package: str = DEFAULT_PACKAGE) -> Optional[RoundCoalescer]:
    """The batcher behind the patched ``compute_modulus`` of `package` (its ``stats`` count launches and requests)."""
    return _round_coalescers.get(package)


def coalescer(package: str = DEFAULT_PACKAGE) -> Optional[Coalescer]:
    """The micro-batcher behind the patched ``_decrypt_raw`` of `package` (its ``stats`` count launches)."""
    return _coalescers.get(package)


def _save(cls: Any, name: str) -> None:
    _saved.setdefault(cls, {})
    if name not in _saved[cls]:
        _saved[cls][name] = cls.__dict__.get(name, None)


# Engine limits that the reference does not have (include/mxpaillier.h); checked once at install time
# and again at the start of compute_modulus, before any message of a keygen round has been exchanged.
MAX_SIEVE_PRIME = (1 << 31) - 1          # mx_sieve: primes < 2^31
MAX_JACOBI_BITS = 257 * 32               # mx_jacobi: moduli up to 257 words (key_length <= 8192, the modexp engine's own limit for N^2)


def check_limits(engine: Any = None, prime_list: Optional[Iterable[int]] = None, prime_length: Optional[int] = None,
                 n_parties: int = 0) -> None:
    """Raises ValueError (before any network round) when a keygen with these parameters would hit an
    engine limit in the middle of ``compute_modulus``."""
    if prime_list is not None:
        top = max(prime_list, default=0)
        if top > MAX_SIEVE_PRIME:
            raise ValueError(
                f"prime_threshold too large for the GPU sieve: largest prime {top} > {MAX_SIEVE_PRIME} "
                "(use prime_threshold < 2^31, or uninstall the patch for this key generation)")
    if prime_length is not None:
        # candidate moduli have 2 * (prime_length + ceil(log2(parties))) bits at most (DK:874-876)
        extra = max(1, (max(1, n_parties) - 1).bit_length())
        bits = 2 * (prime_length + extra)
        if bits > MAX_JACOBI_BITS:
            raise ValueError(
                f"key_length {2 * prime_length} gives candidate moduli of up to {bits} bits; the GPU Jacobi kernel "
                f"takes {MAX_JACOBI_BITS} (key_length <= 8192, the widest key whose N^2 the modexp kernels take)")


def _gpu_key(key: Any, engine: Any) -> GpuPaillierSharedKey:
    cached = getattr(key, "_mx_gpu_key", None)
    if cached is None or cached.n != key.n or cached.share is not key.share:
        # the ciphertext class the key's own module binds (paillier_shared_key.py:16-19 imports it by name)
        import sys

        ct_type = getattr(sys.modules.get(type(key).__module__), "PaillierCiphertext", None)
        cached = GpuPaillierSharedKey.from_reference(key, engine, ciphertext_type=ct_type)
        key._mx_gpu_key = cached
    return cached


def install(engine: Any = None, package: str = DEFAULT_PACKAGE, scalars: bool = True, leaf: bool = False,
            linger: float = 0.0, hw_queues: Optional[int] = None) -> None:
    """``hw_queues``: the engine keeps several launches in flight (chunks of long sequences, the batches of co-located
    parties), which needs more HIP hardware queues than the runtime's default of 4; they can only be chosen before
    the process first touches the GPU.  Default (None): when install() is left to create the engine itself
    (``engine=None`` — the engine is then made on first use, so the runtime is not up yet) it asks for 16 through
    ``configure_hw_queues`` — which sets GPU_MAX_HW_QUEUES in os.environ unless the user set it, a process-wide setting
    that child processes inherit; when the caller passes an engine, the runtime is initialised already and install()
    leaves the environment alone (call ``configure_hw_queues()`` yourself before creating the engine — INTEGRATION.md
    "call order").  An explicit number asks for that many and warns — once per process — when it is too late; 0 never
    touches the environment.  Either way the engine measures what it has and uses no more streams than run side by side.

    ``linger``: seconds a burst of single ``decrypt()`` calls may wait for stragglers before its launch (0 = until
    the event loop has run every coroutine that was runnable, which is what ``asyncio.gather`` over an in-process
    pool needs; a few milliseconds suit pools whose messages arrive over a network).

    ``leaf=True`` additionally rebinds the arithmetic leaf itself — the names ``pow_mod`` / ``mod_inv``
    that distributed_keygen.py:35 and paillier_shared_key.py:20 import from the un-vendored
    tno.mpc.encryption_schemes.utils — to ``operators.pow_mod`` / ``operators.mod_inv``, so that any
    remaining scalar call site of those modules (e.g. ``mod_inv(theta, n)`` in the reference's own
    ``PaillierSharedKey.__init__``, paillier_shared_key.py:50) runs on the engine too.

    ``scalars=False`` leaves the reference's own single-ciphertext ``PaillierSharedKey.partial_decrypt``
    / ``.decrypt`` in place (and with them ``DistributedPaillier.decrypt()`` of ONE ciphertext): a lone
    modexp is a latency-bound chain of ~4800 dependent pair operations — 15-16 ms on the GPU at key_length
    2048 in the two-wavefront latency geometry (39.5 ms in round 2), the same for anything up
    to ~1000 ciphertexts, against ~13 ms for one ``gmpy2.powmod`` on a host core — so deployments that
    decrypt ciphertexts one at a time and care about those milliseconds may prefer the reference's scalar
    path there, while ``decrypt_sequence``, the batch methods and the key generation run on the GPU."""
    psk_mod = importlib.import_module(package + ".paillier_shared_key")
    dk_mod = importlib.import_module(package + ".distributed_keygen")
    PSK = psk_mod.PaillierSharedKey
    DP = dk_mod.DistributedPaillier
    check_limits(engine)
    if hw_queues is None:
        hw_queues = 16 if engine is None else 0
    if hw_queues:
        import os

        from . import configure_hw_queues

        global _warned_late_queues
        if not configure_hw_queues(hw_queues) and int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < hw_queues and not _warned_late_queues:
            import warnings

            _warned_late_queues = True

            warnings.warn(
                "protocols.distributed_keygen_amd.patch.install(): the HIP runtime of this process was initialised before "
                f"the patch was installed, with its default of 4 hardware queues instead of {hw_queues}; launches that are meant "
                "to run side by side (long decrypt_sequence calls, co-located parties) will partly serialise.  Call "
                "patch.install() — or protocols.distributed_keygen_amd.configure_hw_queues() — before the first GPU call.",
                RuntimeWarning, stacklevel=2)
    if leaf:
        from . import operators

        for mod, names in ((psk_mod, ("pow_mod", "mod_inv")), (dk_mod, ("pow_mod", "mod_inv"))):
            for name in names:
                if hasattr(mod, name):
                    _saved_names.setdefault(mod, {}).setdefault(name, getattr(mod, name))
                    op, original = getattr(operators, name), _saved_names[mod][name]

                    def rebound(*a, _op=op, _orig=original):
                        # the engine's arithmetic is Montgomery arithmetic: odd moduli >= 3 (N, N^2 and the
                        # Shamir prime all are).  The leaf the reference imported is total, so anything else
                        # (an even or tiny modulus: no call site of this package produces one) goes to the
                        # function that was bound here before the patch — the reference's own.
                        # int-like arguments (gmpy2.mpz when the reference's utils run on gmpy2, numpy integers) are
                        # coerced once here: the engine's packing takes Python ints.
                        a = tuple(int(v) for v in a)
                        modulus = a[-1]
                        if modulus < 3 or modulus % 2 == 0:
                            return _orig(*a)
                        return _op(*a, engine=engine)

                    setattr(mod, name, rebound)

    # ------------------------------------------------------------------ PaillierSharedKey
    # The scalar methods keep the reference's semantics and hold NO state between calls: any number of
    # decrypt() / decrypt_sequence() coroutines may interleave on one scheme (they have distinct
    # message ids in the reference, DK:352-355 / DK:469-475).
    def partial_decrypt_batch(self: Any, ciphertexts: Iterable[Any]) -> List[int]:
        return _gpu_key(self, engine).partial_decrypt_batch(ciphertexts)

    def decrypt_batch(self: Any, partial_dicts: List[Dict[int, int]]) -> List[int]:
        return _gpu_key(self, engine).decrypt_batch(partial_dicts)

    def partial_decrypt(self: Any, ciphertext: Any) -> int:
        return partial_decrypt_batch(self, [ciphertext])[0]

    def decrypt(self: Any, partial_dict: Dict[int, int]) -> int:
        return decrypt_batch(self, [partial_dict])[0]

    for name, fn in (
        ("partial_decrypt", partial_decrypt),
        ("decrypt", decrypt),
        ("partial_decrypt_batch", partial_decrypt_batch),
        ("decrypt_batch", decrypt_batch),
    ):
        if not scalars and name in ("partial_decrypt", "decrypt"):
            continue
        _save(PSK, name)
        setattr(PSK, name, fn)

    # ------------------------------------------------------------------ decrypt of ONE ciphertext, coalesced
    EncodedPlaintext = dk_mod.EncodedPlaintext
    batcher = _coalescers[package] = Coalescer(engine, linger=linger)

    async def _decrypt_raw(self: Any, ciphertext: Any, receivers: Optional[List[str]] = None):
        """DK:314-382 with the same receivers logic (:331-343), message id (:352-355) and message content (:357-364).
        The partial decryption (:345-349) and the recombination (:378-380) are awaited from the micro-batcher, so
        the coroutines of ``asyncio.gather(*(scheme.decrypt(c) for c in cs))`` share one launch per step instead of
        blocking the loop for one launch each; results and exceptions are per coroutine, as in the reference."""
        if receivers is not None:
            self_receive = "self" in receivers
            receivers_without_self = [recv for recv in receivers if recv != "self"] if self_receive else receivers
        else:
            self_receive = True
            receivers_without_self = receivers
        key = _gpu_key(self.secret_key, engine)
        partial_decryption_shares = {self.index: await batcher.partial_decrypt(key, ciphertext)}
        encryption_hash = bin(ciphertext.peek_value()).zfill(32)[2:34]
        message_id = f"distributed_decryption_session#{self.session_id}_hash#{encryption_hash}"
        if receivers_without_self is None or len(receivers_without_self) != 0:
            self.pool.async_broadcast(
                {"content": "partial_decryption", "value": partial_decryption_shares[self.index]},
                msg_id=message_id,
                handler_names=receivers_without_self,
            )
        if not self_receive:
            return None
        for party, message in await self.pool.recv_all(msg_id=message_id):
            msg_content = message["content"]
            err_msg = f"received a share for {msg_content}, but expected partial_decryption"
            assert msg_content == "partial_decryption", err_msg
            partial_decryption_shares[self.party_indices[party]] = message["value"]
        return EncodedPlaintext(await batcher.decrypt(key, partial_decryption_shares), scheme=self)

    if scalars:
        _save(DP, "_decrypt_raw")
        setattr(DP, "_decrypt_raw", _decrypt_raw)

    # ------------------------------------------------------------------ decrypt_sequence
    _save(DP, "_decrypt_sequence_raw")

    async def _decrypt_sequence_raw(self: Any, ciphertext_sequence: Iterable[Any], receivers: Optional[List[str]] = None):
        """DK:430-517 with the same receivers logic (:447-461), message id (:469-475) and message content
        (:476-483); all state is local to this call."""
        sequence = list(ciphertext_sequence)
        if receivers is not None:
            self_receive = "self" in receivers
            receivers_without_self = [recv for recv in receivers if recv != "self"] if self_receive else receivers
        else:
            self_receive = True
            receivers_without_self = receivers
        key = _gpu_key(self.secret_key, engine)
        # loop DK:463-466 as one launch; the results also stay on the device as this party's column
        partially_decrypted_shares, own_column = key.partial_decrypt_batch(sequence, keep_rows=True)
        encryption_hash = bin(next(iter(sequence)).peek_value()).zfill(32)[2:34] + f"{len(partially_decrypted_shares)}"
        message_id = f"distributed_decryption_session#{self.session_id}_hash#{encryption_hash}"
        if receivers_without_self is None or len(receivers_without_self) != 0:
            self.pool.async_broadcast(
                {"content": "partial_decryption_sequence", "value": partially_decrypted_shares},
                msg_id=message_id,
                handler_names=receivers_without_self,
            )
        if not self_receive:
            return None
        columns: Dict[int, Any] = {self.index: own_column}
        for party, message in await self.pool.recv_all(msg_id=message_id):
            msg_content = message["content"]
            err_msg = f"received a share for {msg_content}, but expected partial_decryption_sequence"
            assert msg_content == "partial_decryption_sequence", err_msg
            columns[self.party_indices[party]] = message["value"]      # ints or wire-form integers, as received
        # loop DK:510-515 as one launch
        plaintexts = key.decrypt_columns(columns, len(sequence))
        return [EncodedPlaintext(m, scheme=self) for m in plaintexts]

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
            try:
                out[i] = [var.get_share(i) for var in batched_v_i.variables]      # every slot set: the usual case
            except KeyError:
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

    rounds_batcher = _round_coalescers[package] = RoundCoalescer(engine, linger=linger)

    async def compute_modulus(
        cls: Any, shares, index, pool, prime_list, party_indices, prime_length, shamir_scheme_t,
        shamir_scheme_2t, correct_param_biprime, session_id, batch_size: int = 1,
    ) -> int:
        check_limits(engine, prime_list, prime_length, len(party_indices))
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
            # DK:1284 (`candidate_n.reconstruct()`) and DK:1288-1292 (the sieve) as one device pass: Lagrange
            # interpolation of all candidates in one launch, rows straight into the sieve, only the
            # survivors' moduli become Python ints
            share_table = [v.get_shares() for v in candidate_n.variables]
            scheme_n = candidate_n.variables[0].shamir_scheme           # degree 2t after the product (UT:248)
            by_party = {i: [tbl[i] for tbl in share_table] for i in share_table[0]}
            # the interpolation points in the order ShamirShares.reconstruct_secret would take them: the first
            # degree+1 entries of the shares dictionary in insertion order
            # (biprime.BiprimeRound: the survivors' moduli and, later, this party's v rows stay on the device between the
            # steps of the round; every value that crosses a communication round does so as a Python int, as in the reference)
            this_round = biprime.BiprimeRound(engine)
            # (awaited from the per-process batcher: parties that share this process and GPU — distributed=False — were
            # handed the same share table and run this step, the v-calculation and the verdicts as ONE launch each)
            surviving = await rounds_batcher.reconstruct_and_sieve(
                this_round, by_party, scheme_n.modulus, scheme_n.polynomial_degree, prime_list,
                points=list(share_table[0])[: scheme_n.polynomial_degree + 1])
            has_divisor = this_round.has_divisor
            survivors = this_round.survivors
            moduli = surviving                                          # candidate index -> modulus, survivors only
            sieved_out += len(has_divisor) - len(survivors)
            if not survivors:
                continue
            g_values = await getattr(cls, mangled + "biprime_test_g_generation")(
                correct_param_biprime, index, [moduli[k] for k in survivors], party_indices, pool,
                f"{sid}_biprime_test_g_{rounds}",
            )
            # DK:1313-1329 as one launch
            v_lists = await rounds_batcher.v_calculation(
                this_round, g_values, index, [p_add[k] for k in survivors], [q_add[k] for k in survivors], correct_param_biprime)
            to_exchange = [
                _to_batched(v, index, moduli[k], correct_param_biprime) for v, k in zip(v_lists, survivors)
            ]
            await exchange_reconstruct(to_exchange, index, pool, party_indices, msg_id=f"{sid}_biprime_test_v_{rounds}_v")
            # DK:1339-1360: slot tests of all survivors as one launch; first passing candidate wins
            verdicts = await rounds_batcher.verdicts(
                this_round, [_v_lists(b, party_indices) for b in to_exchange], correct_param_biprime, errors="return")
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
    for mod, names in _saved_names.items():
        for name, orig in names.items():
            setattr(mod, name, orig)
    _saved_names.clear()
    for cls, names in _saved.items():
        for name, orig in names.items():
            if orig is None:
                if name in cls.__dict__:
                    delattr(cls, name)
            else:
                setattr(cls, name, orig)
    _saved.clear()
    _coalescers.clear()
    _round_coalescers.clear()
