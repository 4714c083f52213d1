#!/usr/bin/env python3
"""Bipartite latency form of the generic modexp: duration of one small launch over the pivot (how many multiplier limbs the
Montgomery wavefront takes; developer knob bi_pivot) — what the library's default pivot is chosen from.
usage: bi_pivot_sweep.py [key_length ...]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L

eng = Engine()
rng = random.Random(5)
LANES = int(os.environ.get("BI_LANES", "0"))        # developer: at least this many lanes per element (knob lat_lanes)
eng.debug_knob("lat_lanes", LANES)
for key_length in [int(a) for a in sys.argv[1:]] or [1024, 2048]:
    bits = key_length + 3
    limbs = L.limbs_for_bits(bits)
    c = 10
    mods = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(c)]
    ebits = bits - 2
    exps = [rng.getrandbits(ebits) | (1 << (ebits - 1)) for _ in mods]
    g = [rng.randrange(m) for m in mods for _ in range(40)]
    g_t = eng.to_device(L.pack(g, limbs))
    mods_t = eng.to_device(L.pack(mods, limbs))
    exps_t = eng.to_device(L.pack(exps, L.limbs_for_bits(ebits)))
    want = [pow(g[k], exps[0], mods[0]) for k in range(3)]

    def run(lpl):
        eng.set_limbs_per_lane(lpl)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = eng.powmod_multi_t(g_t, (mods_t, bits), (exps_t, ebits), 40)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        assert L.unpack(eng.to_host(out[:3])) == want
        return best * 1e3

    print(f"lanes knob {LANES}; geometry {eng.geometry(bits, 40 * c, c)}")
    print(f"key_length {key_length}: {c} candidates x 40 modexps, full-length exponent; one-wavefront latency instance {run(3):.2f} ms")
    eng.set_limbs_per_lane(6)
    steps = 3 * (-(-(bits + 35) // 87)) + 3
    print(f"  default pivot {eng.generic_launch_form(bits, 40 * c, c)[1]} of {steps} steps: {run(6):.2f} ms")
    for pivot in range(3, steps, 3):
        eng.debug_knob("bi_pivot", pivot)
        if eng.generic_launch_form(bits, 40 * c, c)[1] != pivot:
            continue                      # clamped by the library (the factor that leaves the domain must stay below N)
        print(f"  pivot {pivot:3d} (L {pivot} / H {steps - pivot}): {run(6):.2f} ms", flush=True)
    eng.debug_knob("bi_pivot", 0)
