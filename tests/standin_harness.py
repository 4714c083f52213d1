"""Drives the stand-in package (tests/standin/keygen_standin) with three in-process parties over its in-memory
pool — shared by the CPU test (test double of the engine) and the GPU test (the HIP engine) of patch.install()."""

from __future__ import annotations

import asyncio
import importlib
import random
import sys
from pathlib import Path
from typing import Any, Callable, List, Optional

STANDIN_DIR = Path(__file__).resolve().parent / "standin"
PACKAGE = "keygen_standin"
NAMES = ["p1", "p2", "p3"]
if str(STANDIN_DIR) not in sys.path:          # so that patch.install(package=PACKAGE) can import the stand-in
    sys.path.insert(0, str(STANDIN_DIR))


def modules():
    return (importlib.import_module(PACKAGE + ".paillier_shared_key"), importlib.import_module(PACKAGE + ".distributed_keygen"))


class _PublicKey:
    def __init__(self, n: int) -> None:
        self.n = n


class _Scheme:
    def __init__(self, n: int) -> None:
        self.public_key = _PublicKey(n)


def parties_for_key(key: Any, wire: Optional[Callable] = None) -> List[Any]:
    """One DistributedPaillier object per party for a synthetic threshold key (protocols.distributed_keygen_amd
    .synthetic.make_key), wired through one in-memory hub."""
    psk, dk = modules()
    hub = dk.Hub(NAMES, wire=wire)
    out = []
    for i, me in enumerate(NAMES, start=1):
        share = psk.IntegerShares(len(NAMES), {i: key.shares[i]}, key.degree)
        dp = object.__new__(dk.DistributedPaillier)
        dp.secret_key = psk.PaillierSharedKey(n=key.n, t=key.t, player_id=i, share=share, theta=key.theta)
        dp.pool = hub.pool(me)
        dp.index = i
        dp.party_indices = {("self" if n == me else n): k for k, n in enumerate(NAMES, start=1)}
        dp.session_id = 77
        out.append(dp)
    return out


def ciphertexts(key: Any, values: List[int]) -> List[Any]:
    psk, _ = modules()
    scheme = _Scheme(key.n)
    return [psk.PaillierCiphertext(v, scheme) for v in values]


def decrypt_sequence(parties: List[Any], cts: List[Any], return_exceptions: bool = False):
    async def run():
        return await asyncio.gather(*[dp._decrypt_sequence_raw(list(cts)) for dp in parties], return_exceptions=return_exceptions)

    return asyncio.run(run())


def decrypt_single(parties: List[Any], ct: Any):
    async def run():
        return await asyncio.gather(*[dp._decrypt_raw(ct) for dp in parties])

    return asyncio.run(run())


def decrypt_many(parties: List[Any], cts: List[Any], return_exceptions: bool = False):
    """Every party decrypts every ciphertext with a coroutine of its own — `asyncio.gather(*(scheme.decrypt(c) ...))`,
    the shape of the reference's own test (test/test_distributed_keygen.py:132-158).  Results party-major."""
    async def run():
        return await asyncio.gather(*[dp._decrypt_raw(c) for dp in parties for c in cts], return_exceptions=return_exceptions)

    return asyncio.run(run())


def keygen(seed: int, key_length: int, batch_size: int, prime_threshold: int = 200, correct_param: int = 20, t: int = 1) -> List[int]:
    """Three parties run DistributedPaillier.compute_modulus (patched or not) with seeded randomness; returns the
    modulus every party ended with."""
    _, dk = modules()
    dk.rng.seed(seed)
    DP = dk.DistributedPaillier
    hub = dk.Hub(NAMES)
    prime_length, prime_list, sch_t, sch_2t, _ = DP.setup_input(len(NAMES), key_length, prime_threshold, t)

    async def party(i: int, me: str) -> int:
        pool = hub.pool(me)
        party_indices = {("self" if n == me else n): k for k, n in enumerate(NAMES, start=1)}
        return await DP.compute_modulus(dk.Shares(), i, pool, prime_list, party_indices, prime_length, sch_t, sch_2t,
                                        correct_param, 5, batch_size)

    async def run():
        return await asyncio.gather(*[party(i, me) for i, me in enumerate(NAMES, start=1)])

    return asyncio.run(run())
