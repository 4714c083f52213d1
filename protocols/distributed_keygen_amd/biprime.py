"""Host-side mirror of the reference's keygen hot-path class-methods, batched over candidates.

Reference: src/tno/mpc/protocols/distributed_keygen/distributed_keygen.py (DK).

    __small_prime_divisors_test(prime_list, modulus)            DK:1197-1209 -> small_prime_divisors_test[_batch]
    __biprime_test_v_calculation(g_values, index, modulus,
                                 p_i, q_i, correct_param)       DK:1056-1108 -> biprime_test_v_calculation[_batch]
    __biprime_test_with_v_i(batched_v_i, modulus, correct_param,
                            party_indices)                      DK:1110-1175 -> biprime_test_with_v_i[_batch]

The reference wraps the v-values in ``Batched[AdditiveVariable]`` containers (utils.py:301-504);
here they are the plain lists those containers carry, in the same order.  Argument meaning, result
values and error behaviour follow the reference; the batch forms take one entry per candidate
modulus (the loops DK:1288-1292, DK:1313-1329, DK:1339-1360).

The Jacobi-symbol filter of DK:1089 (sympy.jacobi_symbol in the reference) runs on the device too
(``Engine.jacobi_batch``: all generators of all candidates in one launch).
"""

from __future__ import annotations

from typing import Any, Dict, List, Sequence


def _engine(engine: Any) -> Any:
    if engine is not None:
        return engine
    from .engine import default_engine

    return default_engine()


# ------------------------------------------------------------------ DK:1197-1209
def small_prime_divisors_test_batch(prime_list: Sequence[int], moduli: Sequence[int], engine: Any = None) -> List[bool]:
    """[__small_prime_divisors_test(prime_list, n) for n in moduli] (the filter of DK:1288-1292)."""
    if len(moduli) == 0:
        return []
    if len(prime_list) == 0:
        return [False] * len(moduli)
    return _engine(engine).sieve_batch(list(moduli), list(prime_list))


def small_prime_divisors_test(prime_list: Sequence[int], modulus: int, engine: Any = None) -> bool:
    return small_prime_divisors_test_batch(prime_list, [modulus], engine)[0]


# ------------------------------------------------------------------ DK:1056-1108
def biprime_exponent(index: int, modulus: int, p_i: int, q_i: int) -> int:
    """DK:1094 for the party with index 1, DK:1097 for the others."""
    if index == 1:
        return (modulus - p_i - q_i + 1) // 4
    return (p_i + q_i) // 4


def select_generators_batch(
    g_values: Sequence[Sequence[int]], moduli: Sequence[int], correct_param_biprime: int, engine: Any = None
) -> List[List[int]]:
    """Per candidate, the g's DK:1084-1099 exponentiates: in order, Jacobi symbol 1, at most
    correct_param_biprime.  All symbols of all candidates are one GPU launch."""
    symbols = _engine(engine).jacobi_batch(g_values, list(moduli))
    kept: List[List[int]] = []
    for gs, js in zip(g_values, symbols):
        row: List[int] = []
        for g, j in zip(gs, js):
            if len(row) == correct_param_biprime:
                break
            if j == 1:
                row.append(g)
        kept.append(row)
    return kept


def biprime_test_v_calculation_batch(
    g_values: Sequence[Sequence[int]],
    index: int,
    moduli: Sequence[int],
    p_shares: Sequence[int],
    q_shares: Sequence[int],
    correct_param_biprime: int,
    engine: Any = None,
) -> List[List[int]]:
    """One list of v values per candidate (what DK:1103-1107 stores under this party's index).
    All modular exponentiations of all candidates run as ONE GPU launch."""
    if not (len(g_values) == len(moduli) == len(p_shares) == len(q_shares)):
        raise ValueError("one g list, p share and q share per candidate modulus expected")
    if len(moduli) == 0:
        return []
    exps = [biprime_exponent(index, n, p, q) for n, p, q in zip(moduli, p_shares, q_shares)]
    eng = _engine(engine)
    if hasattr(eng, "biprime_v_batch"):
        # Jacobi filter -> selection -> modexps without leaving the device
        return eng.biprime_v_batch(g_values, exps, list(moduli), correct_param_biprime)
    kept = select_generators_batch(g_values, moduli, correct_param_biprime, eng)
    return eng.powmod_batch_multi(kept, exps, list(moduli))


def biprime_test_v_calculation(
    g_values: Sequence[int], index: int, modulus: int, p_i: int, q_i: int, correct_param_biprime: int, engine: Any = None
) -> List[int]:
    return biprime_test_v_calculation_batch([g_values], index, [modulus], [p_i], [q_i], correct_param_biprime, engine)[0]


# ------------------------------------------------------------------ DK:1110-1175
def biprime_test_with_v_i_batch(
    v_by_party: Sequence[Dict[int, Sequence[int]]],
    moduli: Sequence[int],
    correct_param_biprime: int,
    engine: Any = None,
    errors: str = "raise",
) -> List[Any]:
    """Verdict per candidate; ``v_by_party[c][i]`` is the v list of party i for candidate c.
    With ``errors="return"`` a candidate whose test would raise gets the exception object in its
    place (a caller that stops at the first passing candidate, DK:1339-1360, never reaches it).

    Reference semantics: slots are tested in order, False at the first failing slot
    (DK:1160-1164), True after correct_param_biprime passing slots (DK:1168-1172); reaching a
    slot for which a party has no value raises KeyError (utils.py:368-377)."""
    if len(v_by_party) != len(moduli):
        raise ValueError("one v dictionary per candidate modulus expected")
    if len(moduli) == 0:
        return []
    parties = sorted(v_by_party[0].keys())
    if 1 not in parties:
        raise KeyError(1)
    order = [1] + [i for i in parties if i != 1]
    avail = [min(len(v[i]) for i in order) for v in v_by_party]
    nslots = min(correct_param_biprime, max(avail))
    if nslots == 0:
        if correct_param_biprime == 0:
            return [False] * len(moduli)
        if errors == "raise":
            raise KeyError(order[0])
        return [KeyError(order[0]) for _ in moduli]
    # pad short candidates with zeros: their missing slots are never consulted unless reached
    v = [[[int(x) for x in vc[i][:nslots]] + [0] * (nslots - min(nslots, len(vc[i]))) for i in order] for vc in v_by_party]
    slot_pass = _engine(engine).biprime_verdict_batch(v, list(moduli))
    out: List[bool] = []
    for c, passes in enumerate(slot_pass):
        verdict: Any = None
        for k in range(correct_param_biprime):
            if k >= avail[c]:
                # reference: AdditiveVariable.get_share on an unset slot
                if errors == "raise":
                    raise KeyError(order[0])
                verdict = KeyError(order[0])
                break
            if not passes[k]:
                verdict = False
                break
            if k + 1 >= correct_param_biprime:
                verdict = True
                break
        out.append(verdict if isinstance(verdict, Exception) else bool(verdict))
    return out


def biprime_test_with_v_i(
    v_by_party: Dict[int, Sequence[int]], modulus: int, correct_param_biprime: int, engine: Any = None
) -> bool:
    return biprime_test_with_v_i_batch([v_by_party], [modulus], correct_param_biprime, engine)[0]
