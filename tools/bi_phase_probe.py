#!/usr/bin/env python3
"""Where the cycles of a product go in the bipartite form of the generic kernel: one launch of 10 candidates x 40 modexps
per key length on a library built with -DMX_DEV_BI_TRACE (tools/build_variant.py bi_trace -DMX_DEV_BI_TRACE; MX_LIBRARY=...): the
launcher prints the shader-clock cycles pair 0 spent per phase, summed over all products of the exponentiation.
usage: bi_phase_probe.py [key_length ...]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protocols.distributed_keygen_amd import Engine

eng = Engine()
rng = random.Random(3)
for key_length in [int(v) for v in sys.argv[1:]] or [1024, 2048]:
    bits = key_length + 3
    mods = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(10)]
    exps = [rng.getrandbits(bits - 2) | (1 << (bits - 3)) for _ in mods]
    rows = [[rng.randrange(m) for _ in range(40)] for m in mods]
    eng.set_limbs_per_lane(6)
    eng.powmod_batch_multi(rows, exps, mods)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = eng.powmod_batch_multi(rows, exps, mods)
    print(f"key_length {key_length}: int-level call {1e3 * (time.perf_counter() - t0):.2f} ms; products per modexp ~ {bits - 2} squarings + {(bits - 2) // 5 + 31} multiplications (window 5)", flush=True)
    assert got[0][0] == pow(rows[0][0], exps[0], mods[0])
eng.set_limbs_per_lane(0)
