#!/usr/bin/env python3
"""Time-sliced 9-limb launches with the resident workgroups per CU forced (r = 1, 2), a few batch sizes, 2 units per
group — for A/B runs of library variants (MX_LIBRARY)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protocols.distributed_keygen_amd import Engine, _lib, limbs as L, synthetic
if os.environ.get("MX_LIBRARY"):
    probe = ctypes.CDLL(os.environ["MX_LIBRARY"])
    for name in [n for n in _lib.SYMBOLS if not hasattr(probe, n)]:
        del _lib.SYMBOLS[name]
eng = Engine()
key = synthetic.make_key(2048, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp, n = key.exponent(own), key.n
cts = synthetic.random_ciphertexts(key, 12288, seed=7)
c_all = eng.to_device(L.pack(cts, L.limbs_for(key.n_square)))
row = []
for r in (1, 2):
    for b in (4096, 8192, 10000, 12288):
        eng.set_limbs_per_lane(9); eng.set_wavefronts_per_group(2); eng.debug_knob("n2_timeslice", 16 + r)
        eng.powmod_nsquare_t(c_all[:b], n, exp, segments=2); torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); eng.powmod_nsquare_t(c_all[:b], n, exp, segments=2); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        row.append(f"r{r}/b{b} {best * 1e3:.2f}")
print(os.environ.get("MX_LIBRARY", "default").split("/")[-1], " | ".join(row), flush=True)
