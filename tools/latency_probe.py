#!/usr/bin/env python3
"""Latency of small int-level batches (1 .. 4096 ciphertexts, key_length 2048) through
GpuPaillierSharedKey.partial_decrypt_batch, next to gmpy2 / CPython pow on one host core."""
import os, sys, time, subprocess, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protocols.distributed_keygen_amd import Engine, synthetic
from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

eng = Engine()
key = synthetic.make_key(2048, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
gk = GpuPaillierSharedKey(key.n, key.t, own, ShareView(dict(key.shares), key.degree, key.n_fac), key.theta, engine=eng)
cts = synthetic.random_ciphertexts(key, 8192)
for count in (1, 2, 8, 64, 256, 1024, 2048, 4096, 8192):
    best = 1e9
    for rep in range(4):
        batch = [PlainCiphertext(c, key.n) for c in cts[:count]]
        t0 = time.perf_counter()
        out = gk.partial_decrypt_batch(batch)
        best = min(best, time.perf_counter() - t0)
    k, l, w, blk, wv = eng.nsquare_launch_shape(key.n.bit_length(), count)
    print(f"{count:5d} ciphertexts: {best*1e3:7.2f} ms  ({count/best:9.0f} /s)  launch shape K={k} L={l} blocks={blk}, {wv} wavefront(s) per group")
assert out[3] == pow(cts[3], key.exponent(own), key.n_square)
t0 = time.perf_counter()
for c in cts[:8]:
    pow(c, key.exponent(own), key.n_square)
print(f"CPython pow, one core: {(time.perf_counter()-t0)/8*1e3:.1f} ms per modexp")
