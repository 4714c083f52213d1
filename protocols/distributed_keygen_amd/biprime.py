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


def _accepts(fn: Any, keyword: str) -> bool:
    """Does `fn` take `keyword`?  (Feature detection of an engine's device-resident forms: by signature, so that a
    TypeError raised INSIDE a real engine call is never mistaken for a missing feature.)"""
    import inspect

    try:
        params = inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False
    return keyword in params or any(p.kind is inspect.Parameter.VAR_KEYWORD for p in params.values())


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
def _party_column(v_by_party: Sequence[Dict[int, Sequence[int]]], party: int, nslots: int) -> List[int]:
    """All candidates' first `nslots` values of one party as one flat list, short lists padded with zeros (a missing
    slot is never consulted unless the test reaches it, and then it raises)."""
    from itertools import chain

    if all(len(vc[party]) == nslots for vc in v_by_party):
        return list(chain.from_iterable(vc[party] for vc in v_by_party))
    flat: List[int] = []
    for vc in v_by_party:
        vals = vc[party]
        flat.extend(vals[:nslots])
        if len(vals) < nslots:
            flat.extend([0] * (nslots - len(vals)))
    return flat


def biprime_test_with_v_i_batch(
    v_by_party: Sequence[Dict[int, Sequence[int]]],
    moduli: Sequence[int],
    correct_param_biprime: int,
    engine: Any = None,
    errors: str = "raise",
    mods_rows: Any = None,
    own: Any = None,
) -> List[Any]:
    """Verdict per candidate; ``v_by_party[c][i]`` is the v list of party i for candidate c.
    With ``errors="return"`` a candidate whose test would raise gets the exception object in its
    place (a caller that stops at the first passing candidate, DK:1339-1360, never reaches it).

    Reference semantics: slots are tested in order, False at the first failing slot
    (DK:1160-1164), True after correct_param_biprime passing slots (DK:1168-1172); reaching a
    slot for which a party has no value raises KeyError (utils.py:368-377).
    `mods_rows` / `own`: device-resident operands of the same round (BiprimeRound below)."""
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
    eng = _engine(engine)
    if hasattr(eng, "biprime_verdict_columns"):
        # one flat column per party (candidate-major), packed with one codec call each; `own` = (party index, handle of
        # that party's values still on the device): taken from there instead of being packed again
        nested = getattr(eng, "NestedColumn", None)      # one list per candidate, packed as it is (engine.NestedColumn)
        columns = [own[1] if own is not None and own[0] == i else
                   (nested([vc[i] for vc in v_by_party]) if nested is not None else _party_column(v_by_party, i, nslots)) for i in order]
        if _accepts(eng.biprime_verdict_columns, "as_array"):
            slot_pass = eng.biprime_verdict_columns(columns, list(moduli), nslots, mods_rows=mods_rows, as_array=True)
        else:
            slot_pass = eng.biprime_verdict_columns(columns, list(moduli), nslots, mods_rows=mods_rows)
    else:
        # pad short candidates with zeros: their missing slots are never consulted unless reached
        v = [[[int(x) for x in vc[i][:nslots]] + [0] * (nslots - min(nslots, len(vc[i]))) for i in order] for vc in v_by_party]
        slot_pass = eng.biprime_verdict_batch(v, list(moduli))
    # DK:1147-1172 per candidate, slots in order: False at the first failing slot, KeyError on reaching a slot some party
    # has no value for, True after correct_param_biprime passing slots — as array operations (a round has tens of
    # thousands of slots)
    import numpy as np

    passes = np.asarray(slot_pass, dtype=bool).reshape(len(moduli), nslots)
    have = np.minimum(np.asarray(avail, dtype=np.int64), correct_param_biprime)
    consulted = np.arange(nslots, dtype=np.int64)[None, :] < have[:, None]
    fails = (~passes & consulted).any(axis=1)
    missing = ~fails & (have < correct_param_biprime)
    if errors == "raise" and missing.any():
        raise KeyError(order[0])              # reference: AdditiveVariable.get_share on an unset slot
    return [False if f else (KeyError(order[0]) if m else True) for f, m in zip(fails.tolist(), missing.tolist())]


def biprime_test_with_v_i(
    v_by_party: Dict[int, Sequence[int]], modulus: int, correct_param_biprime: int, engine: Any = None
) -> bool:
    return biprime_test_with_v_i_batch([v_by_party], [modulus], correct_param_biprime, engine)[0]


# ------------------------------------------------------------------ one round of DK:1284-1360 with its state on the device
class BiprimeRound:
    """The compute steps of one key-generation round — reconstruction + sieve (DK:1284-1292), this party's v values
    (DK:1313-1329), the verdicts (DK:1339-1360) — with what the later steps need of the earlier ones kept on the device:
    the survivors' moduli (rows straight out of the reconstruction) and this party's v rows.  Between the steps lie the
    reference's communication rounds (generator generation, exchange of the v values), which take and deliver Python
    ints; every value that crosses them is returned / accepted as such, and a value is only taken from the device when
    the caller's copy equals what was computed.  Engines without the device-resident forms (the CPU test double) run
    the same steps through the list-level functions above."""

    def __init__(self, engine: Any = None) -> None:
        self.engine = _engine(engine)
        self.has_divisor: List[bool] = []
        self.survivors: List[int] = []        # candidate indices that passed the sieve, ascending
        self.moduli: List[int] = []           # their moduli, same order
        self._mods_rows: Any = None
        self._own: Any = None                 # (party index, v lists as returned, device handle)

    def reconstruct_and_sieve(self, shares_by_party: Dict[int, Sequence[int]], prime: int, degree: int,
                              prime_list: Sequence[int], points: Any = None) -> Dict[int, int]:
        """DK:1284 + DK:1288-1292 for the whole round; returns {candidate index: modulus} of the survivors."""
        from . import shamir

        eng = self.engine
        pts = shamir._points(shares_by_party, degree, points)
        count = len(shares_by_party[pts[0]])
        if any(len(shares_by_party[i]) != count for i in pts):
            raise ValueError("every party needs one share per candidate")
        self._mods_rows = self._own = None
        if count == 0:
            self.has_divisor, self.survivors, self.moduli = [], [], []
            return {}
        coeffs = shamir.lagrange_coefficients_at_zero(pts, prime, eng)
        columns = [shares_by_party[i] for i in pts]
        if _accepts(eng.shamir_reconstruct_sieve_batch, "keep_rows"):
            self.has_divisor, surviving, self._mods_rows = eng.shamir_reconstruct_sieve_batch(
                columns, coeffs, prime, list(prime_list), keep_rows=True)
        else:                                 # an engine without the device-resident form
            self.has_divisor, surviving = eng.shamir_reconstruct_sieve_batch(columns, coeffs, prime, list(prime_list))
        self.survivors = sorted(surviving)
        self.moduli = [surviving[k] for k in self.survivors]
        return surviving

    # ---- parties that share a process (coalesce.RoundCoalescer) --------------------------------------------------
    def adopt_sieve(self, other: "BiprimeRound") -> Dict[int, int]:
        """The result of `other`.reconstruct_and_sieve for a party that was handed the same share table: the survivors
        (and their moduli on the device, read-only) are shared, nothing is computed."""
        self.has_divisor, self.survivors, self.moduli = list(other.has_divisor), list(other.survivors), list(other.moduli)
        self._mods_rows, self._own = other._mods_rows, None
        return dict(zip(self.survivors, self.moduli))

    def shares_survivors_with(self, other: "BiprimeRound") -> bool:
        return self is other or (self._mods_rows is other._mods_rows and (self.moduli is other.moduli or self.moduli == other.moduli))

    @staticmethod
    def v_calculation_merged(rounds: Sequence["BiprimeRound"], requests: Sequence[Any]) -> List[List[List[int]]]:
        """`rounds[k].v_calculation(*requests[k])` for parties whose rounds hold the SAME survivors and generators, as ONE
        launch: the parties' candidate groups are concatenated — same moduli and generators, every party its own
        exponents (DK:1094 / DK:1097).  requests[k] = (g_values, index, p_shares, q_shares, correct_param_biprime)."""
        first = rounds[0]
        eng, moduli = first.engine, first.moduli
        g_values, keep = requests[0][0], requests[0][4]
        if not moduli:
            return [[] for _ in rounds]
        if first._mods_rows is None or not hasattr(eng, "biprime_verdict_columns") or not _accepts(eng.biprime_v_batch, "mods_rows"):
            return [r.v_calculation(*q) for r, q in zip(rounds, requests)]
        s = len(moduli)
        exps: List[int] = []
        for (gv, index, p_shares, q_shares, kp) in requests:
            if not (len(gv) == s == len(p_shares) == len(q_shares)) or kp != keep:
                raise ValueError("one g list, p share and q share per surviving candidate expected")
            exps.extend(biprime_exponent(index, n, p, q) for n, p, q in zip(moduli, p_shares, q_shares))
        parties = len(rounds)
        lists, rows = eng.biprime_v_batch(list(g_values) * parties, exps, list(moduli) * parties, keep,
                                          mods_rows=first._mods_rows.repeated(parties), keep_rows=True)
        out = []
        for k, (rnd, q) in enumerate(zip(rounds, requests)):
            mine = lists[k * s : (k + 1) * s]
            rnd._own = (q[1], mine, rows.part(k, parties) if rows is not None else None)
            out.append(mine)
        return out

    def v_calculation(self, g_values: Sequence[Sequence[int]], index: int, p_shares: Sequence[int], q_shares: Sequence[int],
                      correct_param_biprime: int) -> List[List[int]]:
        """DK:1313-1329 for the survivors (one g list, p share and q share per survivor, in order)."""
        moduli = self.moduli
        if not (len(g_values) == len(moduli) == len(p_shares) == len(q_shares)):
            raise ValueError("one g list, p share and q share per surviving candidate expected")
        if not moduli:
            return []
        eng = self.engine
        if self._mods_rows is None or not hasattr(eng, "biprime_verdict_columns"):
            return biprime_test_v_calculation_batch(g_values, index, moduli, p_shares, q_shares, correct_param_biprime, eng)
        exps = [biprime_exponent(index, n, p, q) for n, p, q in zip(moduli, p_shares, q_shares)]
        lists, rows = eng.biprime_v_batch(g_values, exps, moduli, correct_param_biprime, mods_rows=self._mods_rows, keep_rows=True)
        self._own = (index, lists, rows)
        return lists

    def verdicts(self, v_by_party: Sequence[Dict[int, Sequence[int]]], correct_param_biprime: int, errors: str = "raise") -> List[Any]:
        """DK:1339-1360 for the survivors: one verdict (or, with errors="return", exception object) per survivor."""
        own = None
        if self._own is not None and self._own[2] is not None and len(v_by_party) == len(self._own[1]):
            index, lists, rows = self._own
            # the device copy stands in for this party's column only if the exchanged values are the computed ones
            if all(index in vc and (vc[index] is mine or list(vc[index]) == mine) for vc, mine in zip(v_by_party, lists)):
                own = (index, rows)
        return biprime_test_with_v_i_batch(v_by_party, self.moduli, correct_param_biprime, self.engine, errors=errors,
                                           mods_rows=self._mods_rows if hasattr(self.engine, "biprime_verdict_columns") else None, own=own)
