"""Python int  <->  little-endian uint32 limb rows (the C ABI's number format).

Row e of a ``[batch, limbs]`` array is ``int.to_bytes(4*limbs, "little")`` of element e.
"""

from __future__ import annotations

from typing import Iterable, List, Sequence

import numpy as np


def limbs_for_bits(bits: int) -> int:
    return max(1, (bits + 31) // 32)


def limbs_for(value: int) -> int:
    return limbs_for_bits(int(value).bit_length())


def pack(values: Sequence[int], limbs: int) -> np.ndarray:
    """ints (0 <= v < 2^(32*limbs)) -> uint32 array [len(values), limbs]."""
    nbytes = 4 * limbs
    try:
        buf = b"".join(int(v).to_bytes(nbytes, "little") for v in values)
    except OverflowError as exc:
        raise ValueError(f"value does not fit in {limbs} uint32 limbs (or is negative)") from exc
    return np.frombuffer(buf, dtype="<u4").reshape(len(values), limbs).copy()


def pack_one(value: int, limbs: int) -> np.ndarray:
    return pack([value], limbs)[0]


def unpack(rows: np.ndarray) -> List[int]:
    """uint32 array [batch, limbs] -> list of ints."""
    rows = np.ascontiguousarray(rows, dtype="<u4")
    if rows.ndim == 1:
        rows = rows.reshape(1, -1)
    nbytes = rows.shape[1] * 4
    raw = rows.tobytes()
    return [int.from_bytes(raw[i * nbytes : (i + 1) * nbytes], "little") for i in range(rows.shape[0])]


def max_bits(values: Iterable[int]) -> int:
    return max((int(v).bit_length() for v in values), default=0)
