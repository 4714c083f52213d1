#!/usr/bin/env python3
"""One decrypt() at key_length 2048 in the latency form of the pair kernel, best and median of N calls (the number the
bench's latency leg reports, without the bench around it) — for A/B runs of library variants through MX_LIBRARY
(tools/build_variant.py).   usage: lone_decrypt_time.py [key_length] [calls] [pivot]   (pivot: developer knob bi_pivot — the
multiplier limbs on the L wavefronts — for sweeps of the split)"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

eng = Engine()
if len(sys.argv) > 3:
    eng.debug_knob("bi_pivot", int(sys.argv[3]))
key = synthetic.make_key(int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 3, 1)
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp, n = key.exponent(own), key.n
cts = synthetic.random_ciphertexts(key, 1, seed=3)
c = eng.to_device(L.pack(cts, L.limbs_for(key.n_square)))
want = pow(cts[0], exp, key.n_square)
times = []
for i in range(calls + 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = eng.powmod_nsquare_t(c, n, exp)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
got = L.unpack(eng.to_host(out))[0]
shape = eng.nsquare_launch_shape(n.bit_length(), 1)
print(f"lone decrypt: best {min(times[2:]):.3f} ms, median {statistics.median(times[2:]):.3f} ms over {calls} calls; "
      f"bit-exact {got == want}; library {os.environ.get('MX_LIBRARY', 'shipped')}; shape {shape}" + (f"; pivot {sys.argv[3]}" if len(sys.argv) > 3 else ""))
