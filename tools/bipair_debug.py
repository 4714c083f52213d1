#!/usr/bin/env python3
"""Stage-by-stage check of the five-wavefront pair kernel against tools/bipair_model.py: runs x^e for a tiny exponent, reads the
pair slots the kernel left in the workspace and compares them, limb for limb and as values, with the model's.
usage: bipair_debug.py [bits] [e]"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

import bipair_model as bp
from bimont_model import Geometry, L, W, limbs_of, value_of
from protocols.distributed_keygen_amd import Engine, limbs as LL

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2053
e = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(11)
n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
n2 = n * n
x = rng.randrange(n2)
geo = Geometry(bits)
cst = bp.PairConstants(n, geo)
cnt = L * geo.K
m, k = W * geo.h_lo, bits - 1
pair_of = lambda v: (limbs_of(v % n2 % n, cnt), limbs_of(v % n2 // n, cnt))
eng = Engine()
eng.set_limbs_per_lane(3)
eng.set_wavefronts_per_group(4)
limbs2 = LL.limbs_for(n2)
c = eng.to_device(LL.pack([x], limbs2))
out = eng.powmod_nsquare_t(c, n, e)
torch.cuda.synchronize()
got = LL.unpack(eng.to_host(out))[0]
print("result ok:", got == pow(x, e, n2))
ws = eng._ws[eng._stream_ptr()].view(torch.int32).cpu().numpy().view(np.uint32)
K = geo.K
nlanes = 2 * 64          # one workgroup of the two-wavefront form = two pair slots
window = eng.nsquare_plan(n, e).desc.window
nslots = 8 + (1 << (window - 1))


def slot(sl, dig):
    return [int(ws[((sl * 2 + dig) * 3 + j) * nlanes + p]) for p in range(K) for j in range(3)]


def show(name, sl, want=None, value=None):
    d0, d1 = slot(sl, 0), slot(sl, 1)
    msg = f"{name:10s}"
    if want is not None:
        msg += f" limbs equal: {d0 == list(want[0])} / {d1 == list(want[1])}"
        if d0 != list(want[0]):
            bad = [i for i in range(cnt) if d0[i] != want[0][i]]
            msg += f" (digit 0 differs at {bad[:6]}: got {[hex(d0[i]) for i in bad[:3]]} want {[hex(want[0][i]) for i in bad[:3]]})"
        if d1 != list(want[1]):
            bad = [i for i in range(cnt) if d1[i] != want[1][i]]
            msg += f" (digit 1 differs at {bad[:6]}: got {[hex(d1[i]) for i in bad[:3]]} want {[hex(want[1][i]) for i in bad[:3]]})"
    if value is not None:
        msg += f" value ok: {bp.pair_value(cst, (d0, d1)) == value % n2}"
    print(msg)
    return d0, d1


K1, K2, ONE = pair_of(1 << (2 * m)), pair_of(1 << (2 * m + k)), pair_of(1 << m)
show("K1", 0, K1)
show("K2", 1, K2)
show("ONE", 3, ONE)
E = ([1 if i == geo.Pd - geo.h_lo else 0 for i in range(cnt)], [0] * cnt)
show("E", 2, E)
xlo, xhi = x & ((1 << k) - 1), x >> k
show("LO", 4, (limbs_of(xlo, cnt), [0] * cnt))
A = bp.pair_mul(geo, cst, (limbs_of(xlo, cnt), [0] * cnt), K1)
show("TMP", 6, A, xlo)
B = bp.pair_mul(geo, cst, (limbs_of(xhi, cnt), [0] * cnt), K2)
S = ([a + b for a, b in zip(A[0], B[0])], [a + b for a, b in zip(A[1], B[1])])
show("T0 (x)", 8, None, x)
if e >= 2:
    show("SQ (x^2)", 7, None, x * x)
show("CARRY/HI", 5, None, pow(x, e, n2) if e else 1)

# ---- the constants of the plan's four-wavefront section against Python
limbs_n = LL.limbs_for(n)
al = lambda v: (v + 255) // 256 * 256
cb = al((8 * limbs_n + 2 * (limbs_n + 1)) * 4)
MAX_OPS = int(os.environ.get("MX_MAX_SLIDING_OPS", "0")) or 16384
sec = 3 * cb + al(MAX_OPS * 4)
blk = eng.nsquare_plan(n, e).block.cpu().numpy().view(np.uint32)
PW = 3 * K
fold_off = (sec + cb) // 4
quot_off = (sec + cb + al(9 * 3 * 64 * 4)) // 4
ok = True
for r in range(7):
    want = limbs_of(pow(2, W * (geo.Pd + r), n), cnt)
    got_r = [int(v) for v in blk[fold_off + r * PW: fold_off + (r + 1) * PW]]
    if got_r != want:
        ok = False
        print("fold row", r, "differs", [i for i in range(cnt) if got_r[i] != want[i]][:5])
    wantq = limbs_of((1 << (W * (geo.Pd + r))) // n, cnt)
    got_q = [int(v) for v in blk[quot_off + r * PW: quot_off + (r + 1) * PW]]
    if got_q != wantq:
        ok = False
        print("quot row", r, "differs", [i for i in range(cnt) if got_q[i] != wantq[i]][:5], [hex(v) for v in got_q[:4]], [hex(v) for v in wantq[:4]])
c2p_w = blk[(sec // 4) + 8 * limbs_n + (limbs_n + 1): (sec // 4) + 8 * limbs_n + 2 * (limbs_n + 1)]
c2p = sum(int(v) << (32 * i) for i, v in enumerate(c2p_w))
print("plan constants ok:", ok, " C2' ok:", c2p == value_of(cst.c2p))
