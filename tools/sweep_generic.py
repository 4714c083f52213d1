#!/usr/bin/env python3
"""Duration of ONE launch of the generic-modulus modexp (per-candidate moduli and exponents, 40 bases per candidate: the
biprimality-test shape, distributed_keygen.py:1084-1099) for every lane geometry over candidate counts — the data the
automatic geometry of mx_powmod_multi_dev is chosen from (profiles/r05_sweep_generic.txt; L3x2 = limbs_per_lane 6, the
bipartite latency form: 3 limbs per lane, every product on two wavefronts).
usage: sweep_generic.py [key_length ...]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L

eng = Engine()
rng = random.Random(5)
for key_length in [int(a) for a in sys.argv[1:]] or [1024, 2048]:
    bits = key_length + 3
    limbs = L.limbs_for_bits(bits)
    counts = (1, 2, 5, 10, 25, 50, 75, 100, 150, 200, 256, 384, 512, 1024, 2048)
    most = max(counts)
    mods = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(most)]
    for what, ebits in (("party 1 (full-length exponent)", bits - 2), ("other parties (half-length exponent)", key_length // 2 + 3)):
        exps = [rng.getrandbits(ebits) | (1 << (ebits - 1)) for _ in mods]
        g = [rng.randrange(m) for m in mods for _ in range(40)]
        g_t = eng.to_device(L.pack(g, limbs))
        mods_t = eng.to_device(L.pack(mods, limbs))
        exps_t = eng.to_device(L.pack(exps, L.limbs_for_bits(ebits)))
        print(f"key_length {key_length}, {what}: ms per launch of candidates x 40 modexps, one launch on an idle GPU")
        print("candidates   L3     L3x2     L9      L18     auto -> (K, L, wavefronts per group)")
        for c in counts:
            row = []
            for lpl in (3, 6, 9, 18, 0):
                eng.set_limbs_per_lane(lpl)
                best = 1e9
                for rep in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    out = eng.powmod_multi_t(g_t[: 40 * c], (mods_t[:c], bits), (exps_t[:c], ebits), 40)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                row.append(best * 1e3)
                if lpl in (3, 6) and c <= 5:      # spot check
                    got = L.unpack(eng.to_host(out[:3]))
                    assert got == [pow(g[k], exps[0], mods[0]) for k in range(3)]
            eng.set_limbs_per_lane(0)
            geo = eng.geometry(bits, 40 * c, c)[:2] + eng.generic_launch_form(bits, 40 * c, c)[:1]
            print(f"{c:9d} {row[0]:7.2f} {row[1]:7.2f} {row[2]:7.2f} {row[3]:7.2f} {row[4]:7.2f}  -> {geo}", flush=True)
