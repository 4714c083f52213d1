"""The reference's arithmetic leaf as operators on the GPU engine.

The reference binds ``pow_mod`` and ``mod_inv`` of the un-vendored tno.mpc.encryption_schemes.utils by
name into the two modules that use them (``from tno.mpc.encryption_schemes.utils import mod_inv,
pow_mod`` at distributed_keygen.py:35 and paillier_shared_key.py:20; call sites
distributed_keygen.py:1094, 1097 and paillier_shared_key.py:50, 90, 92).  This module offers the same
two operators with the same argument order and result — a Python int, the canonical residue — plus
the batched forms the patched call sites use.  A scalar call is a one-element launch (a generic-modulus
modexp; the N^2 partial decryption has its own 15 ms latency path): the scalar forms
exist so that the leaf itself can be rebound (``patch.install(leaf=True)``), the batched forms are what
makes the GPU worthwhile.

Moduli must be odd and >= 3 (N, N^2 and the Shamir prime all are): the engine's arithmetic is
Montgomery arithmetic, and there is no CPU path inside this package to fall through to
(``patch.install(leaf=True)`` sends such moduli to the function the reference had bound before).
"""

from __future__ import annotations

from typing import Any, List, Sequence


def _engine(engine: Any) -> Any:
    if engine is not None:
        return engine
    from .engine import default_engine

    return default_engine()


def pow_mod_batch(values: Sequence[int], exponent: int, modulus: int, engine: Any = None) -> List[int]:
    """[pow_mod(v, exponent, modulus) for v in values]; a negative exponent inverts the values first
    (ValueError if one is not invertible, as ``pow(v, -1, m)``)."""
    eng = _engine(engine)
    values = list(values)
    if exponent < 0:
        values = eng.modinv_batch(values, modulus)
        exponent = -exponent
    return eng.powmod_batch(values, exponent, modulus)


def pow_mod(value: int, exponent: int, modulus: int, engine: Any = None) -> int:
    """``pow_mod(value, exponent, modulus)`` of tno.mpc.encryption_schemes.utils: value**exponent mod modulus."""
    return pow_mod_batch([value], exponent, modulus, engine)[0]


def mod_inv_batch(values: Sequence[int], modulus: int, engine: Any = None) -> List[int]:
    return _engine(engine).modinv_batch(list(values), modulus)


def mod_inv(value: int, modulus: int, engine: Any = None) -> int:
    """``mod_inv(value, modulus)`` of tno.mpc.encryption_schemes.utils; ValueError if gcd(value, modulus) != 1."""
    return mod_inv_batch([value], modulus, engine)[0]


def pow_mod_batch_multi(values: Sequence[Sequence[int]], exponents: Sequence[int], moduli: Sequence[int],
                        engine: Any = None) -> List[List[int]]:
    """[[pow_mod(v, exponents[g], moduli[g]) for v in values[g]] for g]: one (exponent, modulus) per
    group — the shape of the biprimality-test loop distributed_keygen.py:1313-1329."""
    return _engine(engine).powmod_batch_multi([list(v) for v in values], list(exponents), list(moduli))
