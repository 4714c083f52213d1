#!/usr/bin/env python3
"""Column-exact model of the BIPARTITE latency modexp (csrc/mx_bimont.hpp) — developer tool and CPU test vehicle.

Why: a launch of <= ~1000 generic modexps lasts as long as ONE wavefront's dependent chain: a Montgomery
product is word-serial over the limbs of the multiplier, ~16 instructions per limb step on a lone wavefront.  The only way
to shorten that chain is to give each half of the multiplier's limbs to its own wavefront (Kaihara & Takagi's bipartite
modular multiplication):

    a * b * theta  (mod N),   theta = 2^(-W*hL),   b = b_lo + 2^(W*hL) * b_hi
      = [ a * b_lo * 2^(-W*hL) ]      wavefront L: hL least-significant-first Montgomery steps (friendly modulus, as mx_mont.hpp)
      + [ a * b_hi             ]      wavefront H: most-significant-first steps: shift the accumulator up one limb, add
                                      a * b_i, fold what left the top back in through 2^(W*(Ptop+1)) mod N

The model keeps exactly the data the kernel keeps — per lane L = 3 lazy 64-bit columns, W = 29-bit limbs, the same
neighbour exchanges — and asserts every width the kernel relies on (64-bit columns, 32-bit words that cross lanes, the
32-bit fold digit).  tests/test_bimont_model.py runs it against pow().
"""
from __future__ import annotations

W, L = 29, 3
MASK = (1 << W) - 1
U32, U64 = 1 << 32, 1 << 64


class Geometry:
    def __init__(self, nbits: int, h_lo: int = 0) -> None:
        self.nbits = nbits
        # data positions 0 .. Pd-1 hold full limbs, position Pd holds 0 or 1; Pd a multiple of L.  W*Pd >= nbits + 35:
        # the residue of a product is < 2^(W*Pd) * (1 + tiny) + (sum of the final fold digits, < 2^33.6) * N
        self.Pd = L * (-(-(nbits + 35) // (W * L)))
        self.lanes_lo = self.Pd // L + 1                 # wavefront L: positions 0 .. Pd + 2
        self.lanes_hi = self.Pd // L + 2                 # wavefront H: + two product-free lanes that normalise the fold digit
        self.Ptop = L * self.lanes_hi - 1                # = Pd + 5
        k = 1
        while k < self.lanes_hi:
            k <<= 1
        self.K = k
        steps = self.Pd + L                              # multiplier limbs Pd + 2 .. 0 (the top two are zero: alignment to L)
        if not h_lo:
            # (the library's rule, mx_host.hpp choose_geometry: a little more than half of the steps on wavefront L, 0.37 of
            # them for groups of 64 lanes — fitted to tools/bi_pivot_sweep.py)
            h_lo = min(L * ((37 * steps + 150) // (100 * L)) if self.K == 64 else L * ((46 * steps + 600) // (100 * L)), steps - L)
        h_lo = max(h_lo, L, L * (-(-(self.Pd - (nbits - 2) // W) // L)))      # the factor that leaves the domain stays below N
        self.h_lo, self.h_hi = h_lo, steps - h_lo
        assert self.h_lo % L == 0 and self.h_hi % L == 0 and self.h_hi > 0

    def pos_lo(self, p: int, j: int) -> int:
        return L * p + j

    def pos_hi(self, p: int, j: int) -> int:
        return self.Ptop - (L * p + j)


def limbs_of(x: int, count: int):
    out = []
    for _ in range(count):
        out.append(x & MASK)
        x >>= W
    assert x == 0, "value does not fit"
    return out


def value_of(limbs) -> int:
    return sum(v << (W * i) for i, v in enumerate(limbs))


def sq_weight(col_pos: int, i: int) -> int:
    """mx_mont.hpp slot_weight<SQUARE>: cyclic distance of the multiplicand's position from the multiplier limb, mod L"""
    d = (col_pos - i) % L
    if d == 0 or (L % 2 == 0 and d == L // 2):
        return 1
    return 2 if 2 * d < L else 0


class Constants:
    def __init__(self, n: int, geo: Geometry) -> None:
        assert n % 2 == 1 and n >= 3
        self.n, self.geo = n, geo
        u = (-pow(n, -1, 1 << W)) % (1 << W)
        self.nf = limbs_of(u * n + 1, L * geo.K)                         # friendly multiple + 1 (limb 0 is 0)
        assert self.nf[0] == 0
        self.rfold = limbs_of(pow(2, W * (geo.Ptop + 1), n), L * geo.K)  # what one step's overflow is worth
        self.rfin = [limbs_of(pow(2, W * (geo.Pd + k), n), L * geo.K) for k in range(6)]
        self.theta_inv = pow(2, W * geo.h_lo, n)                         # the domain's one
        self.theta = pow(self.theta_inv, -1, n)


def half_lo(geo: Geometry, cst: Constants, a, B, square: bool):
    """Wavefront L: a * (B[0 .. h_lo)) * 2^(-W*h_lo) modulo the friendly multiple; a, B position-indexed lazy limbs.
    Returns almost-normalised limbs (position-indexed, positions 0 .. L*K-1)."""
    K = geo.K
    t = [[0] * L for _ in range(K)]
    al = [[a[geo.pos_lo(p, j)] for j in range(L)] for p in range(K)]
    nf = [[cst.nf[geo.pos_lo(p, j)] for j in range(L)] for p in range(K)]
    for i in range(geo.h_lo):
        bi = B[i]
        for p in range(K):
            for j in range(L):
                w = sq_weight(geo.pos_lo(p, j), i) if square else 1
                t[p][j] += al[p][j] * bi * w
        q = t[0][0] & MASK
        for p in range(K):
            for j in range(L):
                t[p][j] += nf[p][j] * q
                assert t[p][j] < U64
        low = [t[p][0] & MASK for p in range(K)]
        assert low[0] == q and (t[0][0] - q) & MASK == 0
        for p in range(K):
            carry = t[p][0] >> W
            recv = low[p + 1] if p + 1 < K else 0
            t[p] = [t[p][1] + carry, t[p][2], recv]
    # normalize_weak
    out = [0] * (L * K)
    cs = []
    r = [[0] * L for _ in range(K)]
    for p in range(K):
        c = 0
        for j in range(L):
            v = t[p][j] + c
            r[p][j] = v & MASK
            c = v >> W
        cs.append(c)
    for p in range(K):
        cin = cs[p - 1] if p > 0 else 0
        v = r[p][0] + cin
        r[p][0] = v & MASK
        r[p][1] += v >> W
    assert cs[K - 1] == 0
    for p in range(K):
        for j in range(L):
            assert r[p][j] < (1 << W) + (1 << 7)
            out[geo.pos_lo(p, j)] = r[p][j]
    return out


def half_hi_and_sum(geo: Geometry, cst: Constants, a, B, t_lo, square: bool, track=None):
    """Wavefront H: a * (B[h_lo ..]) most significant limb first, then + t_lo, the final fold of the six positions
    Pd .. Pd+5, and the carry sweep.  Returns the product's almost-normalised limbs (position-indexed)."""
    K, Ptop, Pd = geo.K, geo.Ptop, geo.Pd
    nl = geo.lanes_hi

    def at(arr, p, j):
        pos = geo.pos_hi(p, j)
        return arr[pos] if 0 <= pos < len(arr) else 0

    ar = [[at(a, p, j) for j in range(L)] for p in range(K)]
    rf = [[at(cst.rfold, p, j) for j in range(L)] for p in range(K)]
    for p in range(2):                                  # the two normalising lanes take no products and no folds
        assert ar[p] == [0, 0, 0] or (p == 1 and ar[p][0] == ar[p][1] == 0 and ar[p][2] <= 2), ar[p]
        assert rf[p] == [0, 0, 0]
    t = [[0] * L for _ in range(K)]
    vmax = cymax = 0
    # (the limbs at Pd + 1 and Pd + 2 of every multiplier row are zero: the two steps that would open the chain on an empty
    # accumulator are not run)
    assert all(x == 0 for x in B[Pd + 1:]), "a multiplier limb above Pd"
    for i in range(Pd, geo.h_lo - 1, -1):
        bi = B[i] if i < len(B) else 0
        out = [t[p][0] for p in range(K)]
        v = out[0]
        assert v < U32, ("fold digit", v.bit_length())
        vmax = max(vmax, v)
        for p in range(K):
            nxt = out[p + 1] if p + 1 < nl else 0       # lanes beyond the number hold nothing
            cy = nxt >> W
            assert cy < U32, ("carry word", cy.bit_length())
            cymax = max(cymax, cy)
            t[p] = [t[p][1], t[p][2] + cy, nxt & MASK]
        for p in range(K):
            for j in range(L):
                w = sq_weight(geo.pos_hi(p, j), i) if square else 1
                t[p][j] += ar[p][j] * bi * w + rf[p][j] * v
                assert t[p][j] < U64
    def sweep(cols):
        """carry sweep towards the more significant end: column 2 -> 1 -> 0 inside a lane, then ONE hop to the lane above
        (the mirror image of mx_mont.hpp normalize_weak): value preserved, every limb < 2^W + 2^8 afterwards"""
        r = [[0] * L for _ in range(K)]
        cs = []
        for p in range(K):
            c = 0
            for j in (2, 1, 0):
                v = cols[p][j] + c
                if p == 0 and j == 0:           # the number's top column has no lane above it: it keeps its excess
                    r[p][j], c = v, 0
                    assert v < (1 << W) + (1 << 7)
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
            assert max(r[p]) < (1 << W) + (1 << 8)
        return r

    # ---- BEFORE the hand-over (while wavefront L may still be working): sweep this half, so that what stands at
    # positions >= Pd is its true top, and fold the five positions Pd+1 .. Pd+5 (wavefront L's half has nothing there)
    t = sweep(t)
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
    # ---- AFTER the hand-over: + wavefront L's half; position Pd (this half's limb + L's limb there, 0 or 1) is the last
    # digit; fold it, sweep, publish
    for p in range(K):
        for j in range(L):
            t[p][j] += at(t_lo, p, j)
    p0, j0 = 1, L - 1                                   # position Pd = Ptop - 5 lives in lane 1, column 2
    assert geo.pos_hi(p0, j0) == Pd
    digits[0] = t[p0][j0]
    assert digits[0] < (1 << W) + (1 << 9)
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
    assert out[Pd] <= 2, out[Pd]
    if track is not None:
        track["v"] = max(track.get("v", 0), vmax)
        track["cy"] = max(track.get("cy", 0), cymax)
        track["fin"] = max(track.get("fin", 0), max(digits.values()))
    return out


def bmul(geo: Geometry, cst: Constants, a, b, square: bool = False, track=None):
    """a * b * theta mod N on position-indexed lazy limbs (both wavefronts)."""
    t_lo = half_lo(geo, cst, a, b, square)
    return half_hi_and_sum(geo, cst, a, b, t_lo, square, track)


def leave_domain(acc_value: int, n: int, geo: Geometry, reduced: bool = True) -> int:
    """The kernel's epilogue: one plain Montgomery product (radix 2^(W*Pd)) of the accumulator with theta * R =
    2^(W*(Pd - h_lo)), then two conditional subtractions.  `reduced`: the factor modulo n (bisetup_kernel, row 8) — or the
    power of two itself, as the first form of the kernel had it: below a modulus of the LAUNCH's bit length by the
    geometry's clamp, but not below a shorter modulus of another group of the same launch."""
    R = 1 << (W * geo.Pd)
    kout = 1 << (W * (geo.Pd - geo.h_lo))
    if reduced:
        kout %= n
    q = (-acc_value * kout * pow(n, -1, R)) % R
    y = (acc_value * kout + q * n) // R
    assert (acc_value * kout + q * n) % R == 0
    for _ in range(2):
        if y >= n:
            y -= n
    return y


def powmod(g: int, e: int, n: int, win: int = 5, h_lo: int = 0, track=None, launch_bits: int = 0, reduced_exit: bool = True) -> int:
    """g^e mod n the way powmod_bi_kernel computes it: fixed window, every product bipartite.  `launch_bits`: the bit
    length the launch's geometry is derived from (the longest modulus of the launch; default: n's own)."""
    geo = Geometry(max(launch_bits, n.bit_length()), h_lo)
    cst = Constants(n, geo)
    cnt = L * geo.K
    x = limbs_of(g % n * cst.theta_inv % n, cnt)        # into the domain (the kernel: one plain Montgomery product)
    one = limbs_of(cst.theta_inv % n, cnt)
    table = [one, x]
    for _ in range(2, 1 << win):
        table.append(bmul(geo, cst, table[-1], x, False, track))
    ndig = max(1, -(-e.bit_length() // win))
    acc = table[(e >> (win * (ndig - 1))) & ((1 << win) - 1)]
    for d in range(ndig - 2, -1, -1):
        for _ in range(win):
            acc = bmul(geo, cst, acc, acc, True, track)
        acc = bmul(geo, cst, acc, table[(e >> (win * d)) & ((1 << win) - 1)], False, track)
    out = leave_domain(value_of(acc), n, geo, reduced_exit)                # out of the domain
    assert not reduced_exit or out == value_of(acc) * cst.theta % n
    return out


if __name__ == "__main__":
    import random
    import sys

    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    for bits in (1029, 2053, 2050, 131, 515, 4102):
        n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        geo = Geometry(bits)
        tr = {}
        for _ in range(2):
            g, e = rng.randrange(n), rng.getrandbits(64) | 1
            assert powmod(g, e, n, track=tr) == pow(g, e, n)
        print(f"{bits} bits: K {geo.K}, Pd {geo.Pd}, steps L {geo.h_lo} / H {geo.h_hi}; fold digit < 2^{tr['v'].bit_length()}, "
              f"carry word < 2^{tr['cy'].bit_length()}, final digits < 2^{tr['fin'].bit_length()}: ok")
