#!/usr/bin/env python3
"""One line of launch durations (ms) over a fixed set of (key_length, batch, shape) points, for A/B runs of
differently built libraries (MX_LIBRARY=... python tools/variant_probe.py)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import torch
from protocols.distributed_keygen_amd import Engine, _lib, limbs as L, synthetic
if os.environ.get("MX_LIBRARY"):                     # builds of older sources lack the newest entry points
    probe = ctypes.CDLL(os.environ["MX_LIBRARY"])
    for name in [n for n in _lib.SYMBOLS if not hasattr(probe, n)]:
        del _lib.SYMBOLS[name]
eng = Engine()
def knob(name, v):
    try:
        eng.debug_knob(name, v)
    except Exception:
        pass
POINTS = {4096: [(1024, 18, 2), (4096, 18, 2), (2048, 9, 2), (2048, 9, 22), (8192, 18, 1), (1, 3, 2)],
          2048: [(8192, 18, 2), (4096, 9, 2), (4096, 9, 22), (10000, 9, 22), (12288, 9, 2), (32768, 18, 1), (1, 3, 2)]}      # wpg 22: two wavefronts, time-sliced form forced
row = []
for key_length, pts in POINTS.items():
    key = synthetic.make_key(key_length, 3, 1)
    own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
    exp, n = key.exponent(own), key.n
    cts = synthetic.random_ciphertexts(key, max(p[0] for p in pts), seed=7)
    c_all = eng.to_device(L.pack(cts, L.limbs_for(key.n_square)))
    for b, lpl, wpg in pts:
        ts = wpg == 22
        eng.set_limbs_per_lane(lpl); eng.set_wavefronts_per_group(2 if ts else wpg); knob("n2_timeslice", 2 if ts else 1)
        seg = 0 if ts else 1
        eng.powmod_nsquare_t(c_all[:b], n, exp, segments=seg); torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter(); eng.powmod_nsquare_t(c_all[:b], n, exp, segments=seg); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        row.append(f"k{key_length}/b{b}/L{lpl}x{'2ts' if ts else wpg} {best * 1e3:.2f}")
print(os.environ.get("MX_LIBRARY", "default").split("/")[-1], " | ".join(row), flush=True)
