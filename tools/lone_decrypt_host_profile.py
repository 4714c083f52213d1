#!/usr/bin/env python3
"""Where one GpuPaillierSharedKey.partial_decrypt() call (Python int in, Python int out) spends its time beside the kernel:
cProfile of 50 calls at key_length 2048, and the call's time next to the tensor-level launch (tools/lone_decrypt_time.py)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic
from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

eng = Engine()
key = synthetic.make_key(2048, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
gk = GpuPaillierSharedKey(key.n, key.t, own, ShareView(dict(key.shares), key.degree, key.n_fac), key.theta, engine=eng)
cts = synthetic.random_ciphertexts(key, 1, seed=3)
ct = PlainCiphertext(cts[0], key.n)
for _ in range(3):
    gk.partial_decrypt(ct)
times = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    gk.partial_decrypt(ct)
    times.append((time.perf_counter() - t0) * 1e3)
c = eng.to_device(L.pack(cts, L.limbs_for(key.n_square)))
exp = key.exponent(own)
tt = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.powmod_nsquare_t(c, key.n, exp)
    torch.cuda.synchronize()
    tt.append((time.perf_counter() - t0) * 1e3)
print(f"partial_decrypt: best {min(times):.3f} ms median {sorted(times)[10]:.3f} ms; tensor-level launch: best {min(tt):.3f} ms median {sorted(tt)[10]:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    gk.partial_decrypt(ct)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
