#!/usr/bin/env python3
"""When and where every unit of a time-sliced launch ran (developer tool; needs a library built with -DMX_DEV_TS_TRACE:
tools/build_variant.py trace -DMX_DEV_TS_TRACE, then MX_LIBRARY=.../variants/trace.so).
usage: ts_trace.py <batch> <limbs_per_lane 9|18> <resident per CU> <units per group> [key_length]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

b, lpl, r, seg = (int(v) for v in sys.argv[1:5])
key_length = int(sys.argv[5]) if len(sys.argv) > 5 else 2048
eng = Engine()
key = synthetic.make_key(key_length, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp, n = key.exponent(own), key.n
limbs2 = L.limbs_for(key.n_square)
c = eng.to_device(L.pack(synthetic.random_ciphertexts(key, b, seed=7), limbs2))
eng.set_limbs_per_lane(lpl); eng.set_wavefronts_per_group(2); eng.debug_knob("n2_timeslice", 16 + r)
k, l, _w, _blocks, _wf = eng.nsquare_launch_shape(key.n.bit_length(), b)
gpw = 64 // k
groups = (b + gpw - 1) // gpw
nblocks = (b + 2 * gpw - 1) // (2 * gpw)
table_bytes = (72 * 2 * l * nblocks * 128 * 4 + 255) // 256 * 256
for it in range(3):
    out = eng.powmod_nsquare_t(c, n, exp, segments=seg); torch.cuda.synchronize()
ws = eng._ws[eng._stream_ptr()]
off = table_bytes + (32 + groups * 15) * 4
tr = ws[off: off + groups * seg * 16].view(torch.int32).cpu().numpy().astype(np.uint32).reshape(groups * seg, 4)
t0, t1, pair, xcc = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64), tr[:, 2], tr[:, 3] & 0xF
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0                     # microseconds (100 MHz)
dur = t1 - t0
print(f"batch {b}, L{l} x2, K={k}: {groups} groups x {seg} units on {len(set(pair.tolist()))} pairs; launch spans {t1.max() / 1e3:.2f} ms")
for s in range(seg):
    d = dur[s * groups:(s + 1) * groups]
    st = t0[s * groups:(s + 1) * groups]
    print(f"  segment {s}: unit {d.mean() / 1e3:.3f} ms (min {d.min() / 1e3:.3f}, max {d.max() / 1e3:.3f}); starts {st.min() / 1e3:.2f} .. {st.max() / 1e3:.2f} ms")
# per pair: busy time and gaps
order = np.argsort(t0)
busy = {}
gaps = []
last_end = {}
for u in order:
    p = int(pair[u])
    busy[p] = busy.get(p, 0.0) + dur[u]
    if p in last_end:
        gaps.append(t0[u] - last_end[p])
    last_end[p] = t1[u]
gaps = np.array(gaps) if gaps else np.zeros(1)
bt = np.array(list(busy.values()))
print(f"  pairs: busy {bt.mean() / 1e3:.2f} ms on average (min {bt.min() / 1e3:.2f}, max {bt.max() / 1e3:.2f}) of {t1.max() / 1e3:.2f}; "
      f"gap between a pair's units: mean {gaps.mean():.1f} us, p99 {np.percentile(gaps, 99):.1f} us, max {gaps.max():.1f} us")
# hand-overs: who ran segment s+1 of a group
same_pair = same_xcc = total = 0
wait = []
for s in range(seg - 1):
    a, bb = slice(s * groups, (s + 1) * groups), slice((s + 1) * groups, (s + 2) * groups)
    same_pair += int((pair[a] == pair[bb]).sum()); same_xcc += int((xcc[a] == xcc[bb]).sum()); total += groups
    wait.extend((t0[bb] - t1[a]).tolist())
if total:
    wait = np.array(wait)
    print(f"  hand-overs: {total}, to the same pair {same_pair}, within the XCD {same_xcc}; a group waits {wait.mean() / 1e3:.3f} ms between its units (max {wait.max() / 1e3:.2f})")
# occupancy over time
edges = np.linspace(0, t1.max(), 21)
occ = [int(((t0 < e) & (t1 > e)).sum()) for e in edges[1:-1]]
print("  pairs busy at 5 % steps of the launch:", occ)
