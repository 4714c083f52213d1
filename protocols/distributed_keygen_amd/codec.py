"""Wire / disk format of the reference's big integers  <->  limb rows of the C ABI.

The reference serialises with ormsgpack through tno.mpc.communication (options at
distributed_keygen.py:62-68, store/load at distributed_keygen.py:1511-1586).  Python ints that do
not fit msgpack's 64-bit integers travel as ``{"type": "int", "data": <little-endian SIGNED bytes>}``
and registered objects as ``{"type": <class name>, "data": {...}}`` (layout documented in SURVEY.md
§8c from the reference's stored test keys).  Because the payload is already little-endian bytes, a
received list of partial decryptions (distributed_keygen.py:477-505) becomes limb rows by a pad and
a reinterpret — no per-integer Python arithmetic — and a stored key becomes a
``GpuPaillierSharedKey`` directly.  Needs only the stock ``msgpack`` package.
"""

from __future__ import annotations

from typing import Any, Dict, List, Sequence, Tuple

import numpy as np


def encode_int(value: int) -> Dict[str, Any]:
    """Python int -> the reference's msgpack form for big integers."""
    nbytes = (value.bit_length() + 8) // 8  # room for the sign bit
    return {"type": "int", "data": int(value).to_bytes(nbytes, "little", signed=True)}


def decode_int(obj: Any) -> int:
    if isinstance(obj, dict) and obj.get("type") == "int":
        return int.from_bytes(obj["data"], "little", signed=True)
    if isinstance(obj, int):
        return obj
    raise TypeError(f"not a serialised integer: {type(obj)}")


def decode_tree(obj: Any) -> Any:
    """Recursively turn ``{"type","data"}`` wrappers into plain ints / dicts / lists."""
    if isinstance(obj, dict):
        if set(obj.keys()) == {"type", "data"}:
            if obj["type"] == "int":
                return decode_int(obj)
            return decode_tree(obj["data"])
        return {k: decode_tree(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [decode_tree(v) for v in obj]
    return obj


def rows_from_wire(values: Sequence[Any], limbs: int, modulus: int = 0) -> np.ndarray:
    """A received list of big integers (wire form ``{"type": "int", "data": bytes}`` or plain ints) ->
    uint32 rows [len, limbs].  Wire-form entries are copied byte-wise (they ARE little-endian limb
    rows already), plain ints go through the bulk C conversion — no per-integer Python arithmetic on
    the normal path.  With ``modulus`` given, values outside [0, modulus) are reduced the way the
    reference's ``mult_list(..., modulus)`` would reduce them (utils.py:23-38, PSK:115-117);
    without it they raise ValueError."""
    nbytes = 4 * limbs
    if not isinstance(values, (list, tuple)):
        values = list(values)
    from . import limbs as _limbs

    if all(type(v) is int for v in values):
        try:
            rows = _limbs.pack(values, limbs)
        except ValueError:
            if not modulus:
                raise
            return _limbs.pack([v % modulus for v in values], limbs)
        return _reduce_rows(rows, modulus) if modulus else rows
    buf = bytearray(nbytes * len(values))
    for k, v in enumerate(values):
        if isinstance(v, dict) and v.get("type") == "int":
            data = v["data"]
            negative = bool(data) and bool(data[-1] & 0x80)      # signed little-endian: top bit of the last byte
            too_long = len(data) > nbytes and any(data[nbytes:])
            if negative or too_long:
                if not modulus:
                    raise ValueError("negative value" if negative else f"value does not fit in {limbs} uint32 limbs")
                data = (int.from_bytes(data, "little", signed=True) % modulus).to_bytes(nbytes, "little")
            buf[k * nbytes : k * nbytes + min(len(data), nbytes)] = data[:nbytes]
        else:
            iv = int(v)
            if modulus and not 0 <= iv < modulus:
                iv %= modulus
            if iv < 0:
                raise ValueError("negative value")
            try:
                buf[k * nbytes : (k + 1) * nbytes] = iv.to_bytes(nbytes, "little")
            except OverflowError as exc:
                raise ValueError(f"value does not fit in {limbs} uint32 limbs") from exc
    rows = np.frombuffer(bytes(buf), dtype="<u4").reshape(len(values), limbs).copy()
    return _reduce_rows(rows, modulus) if modulus else rows


def _reduce_rows(rows: np.ndarray, modulus: int) -> np.ndarray:
    from . import limbs as _limbs

    return _limbs.reduce_rows(rows, modulus)


def rows_to_wire(rows: np.ndarray) -> List[Dict[str, Any]]:
    """uint32 rows -> list of wire-form integers (minimal signed little-endian bytes)."""
    rows = np.ascontiguousarray(rows, dtype="<u4")
    out = []
    for r in rows:
        raw = r.tobytes().rstrip(b"\x00")
        if not raw or raw[-1] & 0x80:
            raw += b"\x00"
        out.append({"type": "int", "data": raw})
    return out


def load_stored_key(blob: bytes, engine: Any = None) -> Tuple[Any, Dict[str, Any]]:
    """Bytes written by ``DistributedPaillier.store_private_key`` (distributed_keygen.py:1511-1537)
    -> (GpuPaillierSharedKey, metadata dict with public key, precision, index, party_indices,
    corruption_threshold)."""
    import msgpack

    from .shared_key import GpuPaillierSharedKey, ShareView

    obj = decode_tree(msgpack.unpackb(blob, strict_map_key=False)["object"])
    pk = obj["priv_key"]
    sh = pk["share"]
    n_parties = sh["scheme"]["number_of_parties"]
    n_fac = 1
    for k in range(2, n_parties + 1):
        n_fac *= k
    share = ShareView({int(i): v for i, v in sh["shares"].items()}, sh["degree"], n_fac, sh["scaling"])
    key = GpuPaillierSharedKey(n=pk["n"], t=pk["t"], player_id=pk["player_id"], share=share, theta=pk["theta"], engine=engine)
    meta = {k: obj[k] for k in ("pub_key", "precision", "index", "party_indices", "corruption_threshold") if k in obj}
    return key, meta
