"""Host-side mirror of the reference's ``PaillierSharedKey`` whose arithmetic runs on the GPU engine.

Reference: src/tno/mpc/protocols/distributed_keygen/paillier_shared_key.py (PSK).  Same constructor
arguments, attributes, method names, argument meaning and exceptions as PSK:25-127, plus the batched
forms the reference's loops (distributed_keygen.py:463-466 and 510-515) are replaced with:

    partial_decrypt(ciphertext)            PSK:52-93    -> partial_decrypt_batch(ciphertexts)
    decrypt(partial_dict)                  PSK:95-127   -> decrypt_batch(partial_dicts)

``share`` is anything with ``.shares[player_id]``, ``.degree`` and ``.n_fac`` (the reference's
``IntegerShares`` of the un-vendored tno.mpc.encryption_schemes.shamir, or ``ShareView`` below);
a ciphertext is anything with ``.get_value()`` and ``.scheme.public_key.n`` (the reference's
``PaillierCiphertext``; when that class is importable the reference's isinstance check is applied
verbatim).  ``engine`` is injected so the host logic is testable without a GPU; the default is the
process-wide HIP engine; there is no CPU big-integer arithmetic in this module besides the Lagrange
exponent (PSK:70-85: a product of small integers and one exact division) — ``theta_inv`` (PSK:50) is
computed by the engine's device inverse too.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List, Sequence

def _ref_ciphertext_type() -> Any:
    """The reference's ciphertext type, when its (un-vendored) package has been imported in this
    process — a ciphertext of that type cannot exist otherwise, so nothing is imported here."""
    import sys

    mod = sys.modules.get("tno.mpc.encryption_schemes.paillier.paillier")
    return getattr(mod, "PaillierCiphertext", None) if mod is not None else None


@dataclass
class ShareView:
    """The three members of the reference's IntegerShares that PSK:70-85 reads."""

    shares: Dict[int, int]
    degree: int
    n_fac: int
    scaling: int = 0


def _mult_list(values: Iterable[int]) -> int:
    """utils.py:23-38 without modulus."""
    out = 1
    for v in values:
        out *= v
    return out


class GpuPaillierSharedKey:
    """Drop-in for ``PaillierSharedKey`` (PSK:25-127) with batched GPU arithmetic."""

    def __init__(self, n: int, t: int, player_id: int, share: Any, theta: int, engine: Any = None,
                 ciphertext_type: Any = None) -> None:
        # the class PSK:62-65 tests ciphertexts against; None = the reference's PaillierCiphertext when its
        # package has been imported (patch.install passes the one its target package binds)
        self.ciphertext_type = ciphertext_type
        self.share = share
        self.n = n
        self.n_square = n * n
        self.t = t
        self.player_id = player_id
        self.theta = theta
        self._engine = engine
        self.theta_inv = self.engine.modinv_batch([theta], n)[0]  # mod_inv(self.theta, self.n), PSK:50 (ValueError if not invertible)

    @classmethod
    def from_reference(cls, key: Any, engine: Any = None, ciphertext_type: Any = None) -> "GpuPaillierSharedKey":
        """Wrap an existing reference ``PaillierSharedKey`` (same n, t, player_id, share, theta)."""
        return cls(n=key.n, t=key.t, player_id=key.player_id, share=key.share, theta=key.theta, engine=engine,
                   ciphertext_type=ciphertext_type)

    @property
    def engine(self) -> Any:
        if self._engine is None:
            from .engine import default_engine

            self._engine = default_engine()
        return self._engine

    # ------------------------------------------------------------------ PSK:52-93
    def lagrange_exponent(self) -> int:
        """PSK:70-85: n! * prod(other players) * share // prod(j - player_id); may be negative."""
        others = [i + 1 for i in range(self.share.degree + 1) if i + 1 != self.player_id]
        numerator = _mult_list(others)
        denominator = _mult_list([(j - self.player_id) for j in others])
        return (self.share.n_fac * numerator * self.share.shares[self.player_id]) // denominator

    def _check_ciphertext(self, ciphertext: Any, ref_type: Any = None) -> None:
        if ref_type is None:
            ref_type = self.ciphertext_type or _ref_ciphertext_type()
        is_ct = isinstance(ciphertext, PlainCiphertext) or (
            isinstance(ciphertext, ref_type) if ref_type is not None
            else (hasattr(ciphertext, "get_value") and hasattr(ciphertext, "scheme"))
        )
        if not is_ct:  # PSK:62-65
            raise TypeError(f"Expected ciphertext to be a PaillierCiphertext not: {type(ciphertext)}")
        if self.n != ciphertext.scheme.public_key.n:  # PSK:67-68
            raise ValueError("encrypted against a different key!")

    def partial_decrypt_batch(self, ciphertexts: Iterable[Any], keep_rows: bool = False):
        """[self.partial_decrypt(c) for c in ciphertexts] as one GPU batch (DK:463-466).  With
        ``keep_rows`` returns ``(ints, column)``: the column is the engine's device-resident copy of
        the results for ``decrypt_columns`` (the party's own share of the recombination)."""
        values: List[int] = []
        ref_type = self.ciphertext_type or _ref_ciphertext_type()
        ok_type = ok_scheme = None
        for ciphertext in ciphertexts:
            # PSK:62-68 per ciphertext, in order; a type / scheme object that already passed is not re-examined
            if type(ciphertext) is not ok_type or ciphertext.scheme is not ok_scheme:
                self._check_ciphertext(ciphertext, ref_type)
                ok_type, ok_scheme = type(ciphertext), ciphertext.scheme
            values.append(ciphertext.get_value())  # get_value(), not peek_value(): PSK:69
        if not values:
            return ([], None) if keep_rows else []
        exp = self.lagrange_exponent()
        if exp < 0:  # PSK:89-91, as one product-tree inversion on the device
            values = self.engine.modinv_batch(values, self.n_square)
            exp = -exp
        # PSK:92, pow_mod(c, exp, n_square): modulo N^2 through pairs modulo N (mx_powmod_nsquare)
        if keep_rows:
            return self.engine.powmod_nsquare_batch(values, exp, self.n, keep_rows=True)
        return self.engine.powmod_nsquare_batch(values, exp, self.n)

    def partial_decrypt(self, ciphertext: Any) -> int:
        """PSK:52-93."""
        return self.partial_decrypt_batch([ciphertext])[0]

    # ------------------------------------------------------------------ PSK:95-127
    def decrypt_batch(self, partial_dicts: Sequence[Dict[int, int]]) -> List[int]:
        """[self.decrypt(d) for d in partial_dicts] as one GPU batch (DK:510-515).  Raises what
        the reference's loop would raise at the first offending ciphertext."""
        needed = self.share.degree + 1
        rows = []
        for d in partial_dicts:
            rows.append([d[i + 1] for i in range(needed)])  # KeyError if a share is absent, PSK:108-110
        if not rows:
            return []
        messages, ok = self.engine.combine_batch(rows, self.n, self.theta_inv)
        if not all(ok):  # PSK:119-123
            raise ValueError(
                "Combined decryption minus one is not divisible by N. This might be caused by the "
                "fact that the ciphertext that is being decrypted, differs between the parties."
            )
        return messages

    def decrypt(self, partial_dict: Dict[int, int]) -> int:
        """PSK:95-127."""
        return self.decrypt_batch([partial_dict])[0]

    def decrypt_columns(self, columns: Dict[int, Any], count: int) -> List[int]:
        """The loop DK:510-515 over ``count`` ciphertexts with the partial decryptions given per PLAYER
        instead of per ciphertext: ``columns[player]`` is that player's list of partial decryptions
        in ciphertext order as received (ints or wire-form integers, DK:496-505), or the column kept by
        ``partial_decrypt_batch(..., keep_rows=True)`` for this party's own.  Same results and
        exceptions as ``[self.decrypt({p: col[k] for p, col in columns.items()}) for k in range(count)]``:
        KeyError when a needed player (1..degree+1, PSK:108-110) has no value for some ciphertext,
        ValueError when a combination is not 1 modulo N (PSK:119-123)."""
        needed = [i + 1 for i in range(self.share.degree + 1)]
        if count == 0:
            return []
        cols = []
        for player in needed:
            col = columns.get(player)
            if col is None:
                raise KeyError(player)          # shares_dict[player], PSK:110
            length = col.shape[0] if hasattr(col, "shape") else len(col)
            if length < count:                  # the zip of DK:503-505 left later ciphertexts without this share
                raise KeyError(player)
            cols.append(col[:count])
        messages, ok = self.engine.combine_columns(cols, self.n, self.theta_inv)
        if not all(ok):  # PSK:119-123
            raise ValueError(
                "Combined decryption minus one is not divisible by N. This might be caused by the "
                "fact that the ciphertext that is being decrypted, differs between the parties."
            )
        return messages

    # ------------------------------------------------------------------ PSK:186-222
    def __eq__(self, other: object) -> bool:
        if not hasattr(other, "share") or not hasattr(other, "theta"):
            raise TypeError(f"Expected comparison with another PaillierSharedKey, not {type(other)}")
        return (
            self.share == other.share  # type: ignore[attr-defined]
            and self.n == other.n  # type: ignore[attr-defined]
            and self.t == other.t  # type: ignore[attr-defined]
            and self.player_id == other.player_id  # type: ignore[attr-defined]
            and self.theta == other.theta  # type: ignore[attr-defined]
        )

    def __str__(self) -> str:
        return str({"priv_shared_key": {"n": self.n, "t": self.t, "player_id": self.player_id,
                                        "theta": self.theta, "share": self.share}})


class _PlainPublicKey:
    def __init__(self, n: int) -> None:
        self.n, self.g = n, n + 1


class _PlainScheme:
    def __init__(self, n: int) -> None:
        self.public_key = _PlainPublicKey(n)


_plain_schemes: Dict[int, _PlainScheme] = {}


def _plain_scheme(n: int) -> _PlainScheme:
    """One scheme object per public modulus (as the reference's ciphertexts share their scheme)."""
    sch = _plain_schemes.get(n)
    if sch is None:
        if len(_plain_schemes) > 64:
            _plain_schemes.clear()
        sch = _plain_schemes[n] = _PlainScheme(n)
    return sch


@dataclass
class PlainCiphertext:
    """Minimal ciphertext carrier for callers that do not have the tno Paillier package: the raw
    value plus the public modulus, with the reference's ``get_value`` / ``peek_value`` protocol."""

    value: int
    n: int
    fresh: bool = True
    scheme: Any = field(init=False, repr=False)

    def __post_init__(self) -> None:
        self.scheme = _plain_scheme(self.n)

    def get_value(self) -> int:
        self.fresh = False
        return self.value

    def peek_value(self) -> int:
        return self.value
