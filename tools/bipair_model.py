#!/usr/bin/env python3
"""Column-exact model of a BIPARTITE PAIR product — the arithmetic a five-wavefront latency form of the N^2 pair kernel
would run (developer tool, not a kernel's model yet).

The pair kernel (csrc/mx_powmod_n2.hpp) holds x mod N^2 as digits (X0, X1) with x = rho (X0 + X1 N) and multiplies with two
passes modulo N: Z0 = REDC(X0 Y0) together with its exact quotient Q, Z1 = REDC(X0 Y1 + X1 Y0 + C - Q).  Here every pass is
a bipartite product (tools/bimont_model.py: hL least-significant-first friendly Montgomery steps on one wavefront, the other
limbs most-significant-first with folds on a second one), rho = theta = 2^(-W hL):

    pass 1   TL  = (X0 Y0lo + Qm N~) / 2^(W hL)                      digits of Qm recorded       (wavefront AL)
             tH  =  X0 Y0hi - c Vq N,   then the six final folds: - (sum dg_k cf_k) N            (wavefront AH; fold digits recorded)
             Z0  = TL + tH            =>   X0 Y0 = Z0 2^(W hL) - (u Qm - Qc 2^(W hL)) N,   Qc = c Vq + sum dg_k cf_k
    pass 2   TL2 = (X0 Y1lo + X1 Y0lo + [C2 - u Qm] + q' N~) / 2^(W hL)   + Qc                   (wavefront BL)
             tH2 =  X0 Y1hi + X1 Y0hi  (folded)                                                  (wavefront BH)
             Z1  = TL2 + tH2

with c = floor(2^(W (Ptop+1)) / N), cf_k = floor(2^(W (Pd+k)) / N), Vq = sum v_i 2^(W (i - hL)) over the fold digits of the
most-significant-first steps.  The model keeps the kernel's data (L = 3 lazy 64-bit columns per lane, 29-bit limbs, the words
that cross lanes) and asserts every width; run it to see which bounds hold for one and for two product rows."""
from __future__ import annotations

import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from bimont_model import L, MASK, U32, U64, W, Constants, Geometry, limbs_of, sq_weight, value_of  # noqa: E402


def pair_geometry(nbits: int) -> Geometry:
    """The geometry of the pair form (csrc/mx_capi_n2.hip: bipair_geometry): the bipartite geometry with a pivot of its own —
    0.52 of the steps on the L wavefronts, to the nearest block."""
    g = Geometry(nbits)
    steps = g.Pd + L
    h = min(L * ((52 * steps + 150) // (100 * L)), steps - L)
    return Geometry(nbits, h)


class PairConstants(Constants):
    def __init__(self, n: int, geo: Geometry) -> None:
        super().__init__(n, geo)
        self.u = (-pow(n, -1, 1 << W)) % (1 << W)
        cnt = L * geo.K
        self.c = limbs_of((1 << (W * (geo.Ptop + 1))) // n, cnt)                 # quotient of one step's fold
        self.cf = [limbs_of((1 << (W * (geo.Pd + k))) // n, cnt) for k in range(6)]   # ... of the six final folds
        rp = 1 << (W * geo.h_lo)
        c2 = n * (-(-(self.u * (rp - 1)) // n))                                   # C2 = N ceil(u (R' - 1) / N)
        self.c2p = limbs_of(c2 - self.u * (rp - 1), cnt)                           # C2' in [0, N)
        self.theta2 = pow(pow(2, W * geo.h_lo, n * n), -1, n * n)                  # theta modulo N^2


def half_lo(geo, cst, rows, init=None, initq=None, record_q=False):
    """Wavefront L.  rows: [(a, B, square, double)] — sum of a * B[0 .. h_lo) (B doubled limb-wise if `double`) with the
    friendly reduction; init / initq: position-indexed start value init[pos] + u * initq[pos].  Returns (limbs, q digits)."""
    K = geo.K
    pos = lambda p, j: geo.pos_lo(p, j)
    t = [[(init[pos(p, j)] if init else 0) + cst.u * (initq[pos(p, j)] if initq else 0) for j in range(L)] for p in range(K)]
    nf = [[cst.nf[pos(p, j)] for j in range(L)] for p in range(K)]
    qs = []
    for i in range(geo.h_lo):
        for a, B, square, double in rows:
            bi = B[i] << (1 if double else 0)
            for p in range(K):
                for j in range(L):
                    w = sq_weight(pos(p, j), i) if square else 1
                    t[p][j] += a[pos(p, j)] * bi * w
        q = t[0][0] & MASK
        qs.append(q)
        for p in range(K):
            for j in range(L):
                t[p][j] += nf[p][j] * q
                assert t[p][j] < U64, ("L column", t[p][j].bit_length())
        low = [t[p][0] & MASK for p in range(K)]
        assert (t[0][0] - q) & MASK == 0
        for p in range(K):
            carry = t[p][0] >> W
            recv = low[p + 1] if p + 1 < K else 0
            t[p] = [t[p][1] + carry, t[p][2], recv]
    return normalize_weak(geo, t), (qs if record_q else None)


def normalize_weak(geo, t):
    """mx_mont.hpp normalize_weak on position-ordered columns: local sweep, one neighbour exchange, a two-limb fix-up"""
    K = geo.K
    r = [[0] * L for _ in range(K)]
    cs = []
    for p in range(K):
        c = 0
        for j in range(L):
            v = t[p][j] + c
            assert v < U64
            r[p][j] = v & MASK
            c = v >> W
        cs.append(c)
    for p in range(K):
        cin = cs[p - 1] if p > 0 else 0
        v = r[p][0] + cin
        r[p][0] = v & MASK
        r[p][1] += v >> W
    assert cs[K - 1] == 0
    out = [0] * (L * K)
    for p in range(K):
        for j in range(L):
            assert r[p][j] < (1 << W) + (1 << 7), r[p][j].bit_length()
            out[geo.pos_lo(p, j)] = r[p][j]
    return out


def half_hi_and_sum(geo, cst, rows, t_lo, record=None, double_result=False, split_bits=W, track=None):
    """Wavefront H.  rows as in half_lo (limbs h_lo .. Pd+2, most significant first); + t_lo, final folds, sweep.
    record: dict that receives the fold digits {"v": {i: v_i}, "dg": [dg_0 .. dg_5]} (pass 1).  double_result: the half
    is doubled before wavefront L's half is added (a squaring's 2 X0 X1 without a doubled multiplier limb).  split_bits: a
    column that moves to the lane above crosses as its low `split_bits` bits (the new bottom column) and ONE 32-bit word
    for the rest, added with weight 2^(split_bits - W) (the kernel of mx_bimont.hpp: W = 29; a pass with two product rows
    or a doubled multiplier limb accumulates 12 x 2^58 per column and needs 30 to keep the word below 2^32)."""
    K, Ptop, Pd = geo.K, geo.Ptop, geo.Pd
    nl = geo.lanes_hi

    def at(arr, p, j):
        pos = geo.pos_hi(p, j)
        return arr[pos] if 0 <= pos < len(arr) else 0

    ops = [([[at(a, p, j) for j in range(L)] for p in range(K)], B, square, double) for a, B, square, double in rows]
    rf = [[at(cst.rfold, p, j) for j in range(L)] for p in range(K)]
    for ar, _, _, _ in ops:
        for p in range(2):
            assert ar[p] == [0, 0, 0] or (p == 1 and ar[p][0] == ar[p][1] == 0 and ar[p][2] <= 8), ar[p]
    t = [[0] * L for _ in range(K)]
    vmax = cymax = 0
    vs = {}
    # (the limbs at Pd + 1 and Pd + 2 of every multiplier row are zero — a product leaves at most 4 at Pd and nothing above,
    # asserted below; constants and the halves of x end below Pd — so the two steps that would open the chain on an empty
    # accumulator are not run: their fold digits stay the zeros the V buffer was cleared to)
    for _, B, _, _ in rows:
        assert all(x == 0 for x in B[Pd + 1:]), "a multiplier limb above Pd"
    for i in range(Pd, geo.h_lo - 1, -1):
        out = [t[p][0] for p in range(K)]
        v = out[0]
        assert v < (1 << split_bits) + 64, ("fold digit", v.bit_length(), v - (1 << split_bits))
        vs[i] = v
        vmax = max(vmax, v)
        for p in range(K):
            nxt = out[p + 1] if p + 1 < nl else 0
            cy = nxt >> split_bits
            assert cy < U32, ("carry word", cy.bit_length())
            cymax = max(cymax, cy)
            t[p] = [t[p][1], t[p][2] + (cy << (split_bits - W)), nxt & ((1 << split_bits) - 1)]
        for ar, B, square, double in ops:
            bi = (B[i] if i < len(B) else 0) << (1 if double else 0)
            for p in range(K):
                for j in range(L):
                    w = sq_weight(geo.pos_hi(p, j), i) if square else 1
                    t[p][j] += ar[p][j] * bi * w
        for p in range(K):
            for j in range(L):
                t[p][j] += rf[p][j] * v
                assert t[p][j] < U64

    def sweep(cols, top_bound=(1 << W) + (1 << 7)):
        r = [[0] * L for _ in range(K)]
        cs = []
        for p in range(K):
            c = 0
            for j in (2, 1, 0):
                v = cols[p][j] + c
                if p == 0 and j == 0:
                    r[p][j], c = v, 0
                    assert v < top_bound, v.bit_length()
                else:
                    r[p][j] = v & MASK
                    c = v >> W
            assert c < (1 << 40)
            cs.append(c)
        for p in range(K):
            cin = cs[p + 1] if p + 1 < K else 0
            v = r[p][2] + cin
            r[p][2] = v & MASK
            r[p][1] += v >> W
            assert max(r[p][1:] if p == 0 else r[p]) < (1 << W) + (1 << 8) and r[p][0] < top_bound + (1 << 8)
        return r

    t = sweep(t, top_bound=(1 << split_bits) + (1 << 9))
    if double_result:
        t = sweep([[2 * x for x in lane] for lane in t], top_bound=(1 << (W + 1)) + (1 << 9))
    fin = [[[at(cst.rfin[k], p, j) for j in range(L)] for p in range(K)] for k in range(6)]
    digits = {}
    for p in range(2):
        for j in range(L):
            k = geo.pos_hi(p, j) - Pd
            assert 0 <= k < 6
            if k >= 1:
                digits[k] = t[p][j]
                t[p][j] = 0
    for p in range(K):
        for j in range(L):
            for k in range(1, 6):
                t[p][j] += fin[k][p][j] * digits[k]
    for p in range(K):
        for j in range(L):
            t[p][j] += at(t_lo, p, j)
    p0, j0 = 1, L - 1
    assert geo.pos_hi(p0, j0) == Pd
    digits[0] = t[p0][j0]
    assert digits[0] < (1 << W) + (1 << 10), digits[0].bit_length()
    t[p0][j0] = 0
    for p in range(K):
        for j in range(L):
            t[p][j] += fin[0][p][j] * digits[0]
            assert t[p][j] < U64, t[p][j].bit_length()
    r = sweep(t)
    out = [0] * (L * K)
    for p in range(K):
        for j in range(L):
            pos = geo.pos_hi(p, j)
            if pos >= 0:
                assert r[p][j] < (1 << W) + (1 << 7)
                out[pos] = r[p][j]
            else:
                assert r[p][j] == 0
    for pos in range(Pd + 1, L * K):
        assert out[pos] == 0, (pos, out[pos])
    assert out[Pd] <= 4, out[Pd]
    if record is not None:
        record["v"] = vs
        record["dg"] = [digits[k] for k in range(6)]
    if track is not None:
        track["v"] = max(track.get("v", 0), vmax)
        track["cy"] = max(track.get("cy", 0), cymax)
        track["fin"] = max(track.get("fin", 0), max(digits.values()))
        track["top"] = max(track.get("top", 0), out[Pd])
    return out


def quotient_of_the_high_half(geo, cst, rec):
    """Qc = c * Vq + sum dg_k * cf_k as wavefront AL would form it: for each of its positions the lazy column
    sum_{i} c_i * V[pos - i] + sum_k dg_k cf_k[pos] (asserted < 2^64), then normalize_weak."""
    K = geo.K
    V = [0] * (L * K)
    for i, v in rec["v"].items():
        V[i - geo.h_lo] = v
    nc = max(k for k, x in enumerate(cst.c) if x) + 1
    t = [[0] * L for _ in range(K)]
    for p in range(K):
        for j in range(L):
            pos = geo.pos_lo(p, j)
            col = sum(cst.c[i] * V[pos - i] for i in range(nc) if 0 <= pos - i < len(V))
            col += sum(rec["dg"][k] * cst.cf[k][pos] for k in range(6))
            assert col < U64, col.bit_length()
            t[p][j] = col
    return normalize_weak(geo, t), nc


def pair_mul(geo, cst, X, Y, square=False, track=None, sq_double_result=False, split2=W + 1):
    """(Z0, Z1) with theta (Z0 + Z1 N) = [theta (X0 + X1 N)] [theta (Y0 + Y1 N)] modulo N^2; square: Y is X."""
    X0, X1 = X
    Y0, Y1 = (X0, X1) if square else Y
    # ---- pass 1
    tl, qm = half_lo(geo, cst, [(X0, Y0, square, False)], record_q=True)
    rec = {}
    z0 = half_hi_and_sum(geo, cst, [(X0, Y0, square, False)], tl, record=rec, track=track)
    qc, nc = quotient_of_the_high_half(geo, cst, rec)
    if track is not None:
        track["c_limbs"] = nc
    # ---- pass 2: start value C2' + u (2^(W hL) - 1 - Qm) limb-wise, Qc added to wavefront L's half
    cnt = L * geo.K
    initq = [(MASK - qm[i]) if i < geo.h_lo else 0 for i in range(cnt)]
    if square:
        rows_lo = [(X0, X1, False, True)]                              # 2 X0 X1: the multiplier limb doubled (F_BDOUBLE)
        rows_hi = [(X0, X1, False, not sq_double_result)]
    else:
        rows_lo = rows_hi = [(X0, Y1, False, False), (X1, Y0, False, False)]
    tl2, _ = half_lo(geo, cst, rows_lo, init=cst.c2p, initq=initq)
    tl2 = [a + b for a, b in zip(tl2, qc)]
    tr2 = {} if track is not None else None
    z1 = half_hi_and_sum(geo, cst, rows_hi, tl2, double_result=square and sq_double_result, split_bits=split2, track=tr2)
    if track is not None:
        for k, v in tr2.items():
            track[k + "2"] = max(track.get(k + "2", 0), v)
    return z0, z1


def pair_value(cst, Z) -> int:
    n2 = cst.n * cst.n
    return cst.theta2 * (value_of(Z[0]) + value_of(Z[1]) * cst.n) % n2


def to_pair(geo, cst, x: int):
    """digits of x / theta modulo N^2 (what the conversion products of the kernel deliver), exact limbs"""
    n = cst.n
    s = x * pow(cst.theta2, -1, n * n) % (n * n)
    cnt = L * geo.K
    return limbs_of(s % n, cnt), limbs_of(s // n, cnt)


if __name__ == "__main__":
    import random

    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    for bits in (2053, 2051, 1029, 515, 4102):
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        geo = Geometry(bits)
        cst = PairConstants(n, geo)
        n2 = n * n
        tr = {}
        x = rng.randrange(n2)
        X = to_pair(geo, cst, x)
        acc, val = X, x
        for step in range(12):
            if step % 4 == 3:
                acc, val = pair_mul(geo, cst, acc, X, False, tr), val * x % n2
            else:
                acc, val = pair_mul(geo, cst, acc, acc, True, tr), val * val % n2
            assert pair_value(cst, acc) == val, (bits, step)
        print(f"{bits} bits: K {geo.K}, Pd {geo.Pd}, steps L {geo.h_lo} / H {geo.h_hi}, c has {tr['c_limbs']} limbs; pass 1: fold digit < 2^{tr['v'].bit_length()}, "
              f"carry word < 2^{tr['cy'].bit_length()}, final digits < 2^{tr['fin'].bit_length()}, limb Pd <= {tr['top']}; pass 2: < 2^{tr['v2'].bit_length()}, "
              f"< 2^{tr['cy2'].bit_length()}, < 2^{tr['fin2'].bit_length()}, <= {tr['top2']}: ok")
