"""Host-side mirror of the Shamir-field steps of a key-generation round, batched over candidates.

Reference: the candidate moduli of a round are formed from Shamir shares modulo a prime P
(distributed_keygen.py:647-651) by

    candidate_n = prime_candidate_p * prime_candidate_q         DK:1274   share-wise product modulo P
    candidate_n += zero                                         DK:1277   share-wise sum modulo P
    await exchange_reconstruct(candidate_n, ...)                DK:1281   every party learns all shares
    candidate_n_plaintext = candidate_n.reconstruct()           DK:1284   Lagrange interpolation at 0

through ``Batched[ShamirVariable]`` (utils.py:205-270, 404-471), whose arithmetic lives in the
un-vendored tno.mpc.encryption_schemes.shamir (``ShamirShares.__mul__/__add__/reconstruct_secret``:
the textbook prime-field operations).  Here the same values are computed for the whole batch at
once: ``mul_add_shares_batch`` = this party's share of every candidate, ``reconstruct_batch`` = the
candidate moduli; the tensor-level forms (``Engine.shamir_fma_t`` / ``shamir_lincomb_t``) keep them
on the device for the sieve that follows (DK:1288-1292).
"""

from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence


def _engine(engine: Any) -> Any:
    if engine is not None:
        return engine
    from .engine import default_engine

    return default_engine()


def lagrange_coefficients_at_zero(points: Sequence[int], prime: int, engine: Any = None) -> List[int]:
    """lambda_i = prod_{j != i} x_j / (x_j - x_i) mod prime for the evaluation points x (party indices):
    products of a handful of word-sized values on the host, as in the reference; the modular inverses
    of the denominators go through the engine's device inverse when an engine is given (so that the
    batched path has no host big-integer inversion at all), else through ``pow(den, -1, prime)``."""
    if len(set(points)) != len(points):
        raise ValueError("evaluation points must be distinct")
    nums, dens = [], []
    for i in points:
        num = den = 1
        for j in points:
            if j != i:
                num = num * j % prime
                den = den * (j - i) % prime
        nums.append(num)
        dens.append(den)
    invs = engine.modinv_batch(dens, prime) if engine is not None else [pow(d, -1, prime) for d in dens]
    return [n * v % prime for n, v in zip(nums, invs)]


def mul_add_shares_batch(p_shares: Sequence[int], q_shares: Sequence[int], zero_shares: Sequence[int], prime: int,
                         engine: Any = None) -> List[int]:
    """[(p * q + z) % prime ...]: this party's share of every candidate modulus (DK:1274-1277)."""
    return _engine(engine).shamir_fma_batch(list(p_shares), list(q_shares), list(zero_shares), prime)


def _points(shares_by_party: Dict[int, Sequence[int]], degree: int, points: Optional[Sequence[int]]) -> List[int]:
    """The degree+1 evaluation points the interpolation runs through.  The un-vendored
    ``ShamirShares.reconstruct_secret`` takes the first degree+1 entries of its shares dictionary in
    INSERTION order; a caller that has that dictionary passes its key order as `points` (patch.py does).
    Without it the parties are taken in index order — the same value whenever the shares are consistent
    (any degree+1 points of a degree-`degree` polynomial interpolate to the same secret)."""
    if points is None:
        pts = sorted(shares_by_party)[: degree + 1]
    else:
        pts = [int(i) for i in points][: degree + 1]
        if any(i not in shares_by_party for i in pts):
            raise KeyError(next(i for i in pts if i not in shares_by_party))
    if len(pts) < degree + 1:
        raise ValueError("not enough shares to reconstruct")
    return pts


def reconstruct_batch(shares_by_party: Dict[int, Sequence[int]], prime: int, degree: int, engine: Any = None,
                      points: Optional[Sequence[int]] = None) -> List[int]:
    """Candidate moduli of a round from every party's shares (DK:1284): per candidate the value at 0 of
    the degree-`degree` polynomial through the shares of degree+1 parties (`points`, see _points).
    Raises ValueError with fewer shares than that."""
    points = _points(shares_by_party, degree, points)
    count = len(shares_by_party[points[0]])
    if any(len(shares_by_party[i]) != count for i in points):
        raise ValueError("every party needs one share per candidate")
    if count == 0:
        return []
    eng = _engine(engine)
    coeffs = lagrange_coefficients_at_zero(points, prime, eng)
    return eng.shamir_lincomb_batch([shares_by_party[i] for i in points], coeffs, prime)


def reconstruct_and_sieve_batch(shares_by_party: Dict[int, Sequence[int]], prime: int, degree: int,
                                prime_list: Sequence[int], engine: Any = None, points: Optional[Sequence[int]] = None):
    """DK:1284 and the filter DK:1288-1292 for a whole round without the moduli leaving the device in
    between: returns (has_small_divisor per candidate, {candidate index: modulus} of the survivors)."""
    points = _points(shares_by_party, degree, points)
    count = len(shares_by_party[points[0]])
    if any(len(shares_by_party[i]) != count for i in points):
        raise ValueError("every party needs one share per candidate")
    if count == 0:
        return [], {}
    eng = _engine(engine)
    coeffs = lagrange_coefficients_at_zero(points, prime, eng)
    return eng.shamir_reconstruct_sieve_batch([shares_by_party[i] for i in points], coeffs, prime, list(prime_list))
