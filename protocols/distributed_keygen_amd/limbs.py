"""Python int  <->  little-endian uint32 limb rows (the C ABI's number format).

Row e of a ``[batch, limbs]`` array is ``int.to_bytes(4*limbs, "little")`` of element e.
"""

from __future__ import annotations

import operator
from typing import Iterable, List, Sequence

import numpy as np


def limbs_for_bits(bits: int) -> int:
    return max(1, (bits + 31) // 32)


def limbs_for(value: int) -> int:
    return limbs_for_bits(int(value).bit_length())


try:  # bulk conversion in C (csrc/mx_pycodec.c, built by build.py next to the HIP library)
    from . import _mxcodec  # type: ignore
except ImportError:  # pragma: no cover - a checkout that has not been built yet
    _mxcodec = None


def _codec():
    """The C helper, looked up again if this module was imported before build.py had produced it."""
    global _mxcodec
    if _mxcodec is None:
        try:
            from . import _mxcodec as mod  # type: ignore

            _mxcodec = mod
        except ImportError:
            pass
    return _mxcodec


def as_index(value):
    """int(value) if `value` is an integer scalar (int, gmpy2.mpz, numpy integer ...: anything with __index__ that is
    not a sequence), else None — how the functions below tell ONE modulus from a sequence of moduli."""
    if isinstance(value, int):
        return int(value)
    if hasattr(value, "__len__") or hasattr(value, "__iter__"):
        return None
    try:
        return int(operator.index(value))
    except TypeError:
        try:                                        # gmpy2.mpz before 2.1 has __int__ only
            return int(value) if type(value).__name__ == "mpz" else None
        except (TypeError, ValueError):
            return None


def pack_into(values: Sequence[int], limbs: int, out: np.ndarray, row_offset: int = 0) -> None:
    """ints -> rows [row_offset, row_offset + len(values)) of the C-contiguous uint32 array `out`
    (e.g. a pinned staging buffer), without intermediate copies."""
    codec = _codec()
    if codec is not None and isinstance(values, (list, tuple)) and all(type(v) is int for v in values[:1]):
        try:
            codec.pack_into(values, limbs, out, row_offset)
            return
        except TypeError:
            pass                                    # int-like objects (e.g. gmpy2.mpz): the generic path converts them
    nbytes = 4 * limbs
    flat = out.reshape(-1).view(np.uint8)
    try:
        for k, v in enumerate(values):
            flat[(row_offset + k) * nbytes : (row_offset + k + 1) * nbytes] = np.frombuffer(int(v).to_bytes(nbytes, "little"), dtype=np.uint8)
    except OverflowError as exc:
        raise ValueError(f"value does not fit in {limbs} uint32 limbs (or is negative)") from exc


def pack_nested_into(lists: Sequence[Sequence[int]], inner: int, limbs: int, out: np.ndarray, row_offset: int = 0) -> None:
    """lists[g] -> rows [row_offset + g * inner, row_offset + (g + 1) * inner) of `out`: the first min(len, inner) ints of
    the list, then zero rows — one list per candidate as the reference holds them (the generator lists of
    distributed_keygen.py:1313-1329, a party's v lists of :1339-1360), packed without flattening them into one list of
    references first (230 000 of them in a 65 536-candidate round)."""
    codec = _codec()
    if codec is not None and hasattr(codec, "pack_nested_into") and isinstance(lists, (list, tuple)):
        try:
            codec.pack_nested_into(lists, inner, limbs, out, row_offset)
            return
        except TypeError:
            pass                                    # int-like elements: the generic path converts them
    flat: List[int] = []
    for vals in lists:
        vals = list(vals)[:inner]
        flat.extend(vals)
        flat.extend([0] * (inner - len(vals)))
    pack_into(flat, limbs, out, row_offset)


def unpack_groups(rows: np.ndarray, counts: Sequence[int], stride: int) -> List[List[int]]:
    """uint32 array [groups * stride, limbs] -> per group the ints of its first counts[g] rows, as lists built in one
    pass (the v lists of distributed_keygen.py:1103-1108; rows behind a group's count are never turned into ints)."""
    rows = np.ascontiguousarray(rows, dtype="<u4")
    counts = [int(c) for c in counts]
    codec = _codec()
    if codec is not None and hasattr(codec, "unpack_groups"):
        return codec.unpack_groups(rows, rows.shape[1], counts, stride) if counts else []
    vals = unpack(rows)
    if any(c < 0 or c > stride for c in counts) or len(counts) * stride > len(vals):
        raise ValueError("a count must lie in 0 .. stride and the rows must hold groups x stride")
    return [vals[g * stride : g * stride + c] for g, c in enumerate(counts)]


def pack(values: Sequence[int], limbs: int) -> np.ndarray:
    """ints (0 <= v < 2^(32*limbs)) -> uint32 array [len(values), limbs]."""
    if not isinstance(values, (list, tuple)):
        values = list(values)
    out = np.empty((len(values), limbs), dtype="<u4")
    pack_into(values, limbs, out, 0)
    return out


def reduce_rows(rows: np.ndarray, moduli) -> np.ndarray:
    """Rows (uint32 [count, limbs]) that are >= their modulus -> their residue, in place.  `moduli`: one int for
    all rows, or a sequence of ints with count = len(moduli) * group (consecutive rows share a modulus).
    Vectorised: the word that holds a modulus' top bit decides for all but a handful of rows (a value is below
    the modulus when that word is smaller and nothing lies above it); only rows that may be >= it become Python
    ints.  This replaces a Python-level ``v % m`` per element on the int-level paths, where received values are
    canonical residues already."""
    count, limbs = rows.shape
    if count == 0:
        return rows
    one = as_index(moduli)
    mods = [one] if one is not None else [int(m) for m in moduli]
    group = count // len(mods)
    codec = _codec()
    room = 32 * limbs
    if codec is not None and hasattr(codec, "rows_ge") and rows.flags.c_contiguous and all(m > 0 and m.bit_length() <= room for m in mods):
        # one C pass compares every row with its modulus word by word; normally nothing comes back
        for k in codec.rows_ge(rows, limbs, pack(mods, limbs), group):
            v = int.from_bytes(rows[k].tobytes(), "little")
            rows[k] = np.frombuffer((v % mods[k // group]).to_bytes(4 * limbs, "little"), dtype="<u4")
        return rows
    view = rows.reshape(len(mods), group, limbs)
    for g, m in enumerate(mods):
        top = (m.bit_length() - 1) // 32
        if top >= limbs:
            continue                                  # every value of this width is below the modulus
        blk = view[g]
        suspect = blk[:, top] >= np.uint32((m >> (32 * top)) & 0xFFFFFFFF)
        if top + 1 < limbs:
            suspect |= blk[:, top + 1 :].any(axis=1)
        for k in np.nonzero(suspect)[0]:
            v = int.from_bytes(blk[k].tobytes(), "little")
            if v >= m:
                blk[k] = np.frombuffer((v % m).to_bytes(4 * limbs, "little"), dtype="<u4")
    return rows


def pack_reduced(values: Sequence[int], limbs: int, moduli) -> np.ndarray:
    """ints -> uint32 rows of their residues modulo `moduli` (one modulus, or one per group of consecutive
    values): the bulk C conversion plus reduce_rows; values that do not fit the rows or are negative take the
    per-element path."""
    if not isinstance(values, (list, tuple)):
        values = list(values)
    try:
        rows = pack(values, limbs)
    except ValueError:
        one = as_index(moduli)
        if one is not None:
            return pack([int(v) % one for v in values], limbs)
        mods = [int(m) for m in moduli]
        group = len(values) // len(mods)
        return pack([int(v) % mods[k // group] for k, v in enumerate(values)], limbs)
    return reduce_rows(rows, moduli)


def pack_one(value: int, limbs: int) -> np.ndarray:
    return pack([value], limbs)[0]


def unpack(rows: np.ndarray) -> List[int]:
    """uint32 array [batch, limbs] -> list of ints."""
    rows = np.ascontiguousarray(rows, dtype="<u4")
    if rows.ndim == 1:
        rows = rows.reshape(1, -1)
    codec = _codec()
    if codec is not None:
        return codec.unpack(rows, rows.shape[1]) if rows.shape[0] else []
    nbytes = rows.shape[1] * 4
    raw = rows.tobytes()
    return [int.from_bytes(raw[i * nbytes : (i + 1) * nbytes], "little") for i in range(rows.shape[0])]


def max_bits(values: Iterable[int]) -> int:
    return max((int(v).bit_length() for v in values), default=0)
