#!/usr/bin/env python3
"""One decrypt() at key_length 2048 on a library built with -DMX_DEV_BP_TRACE (tools/build_variant.py bp_trace -DMX_DEV_BP_TRACE;
MX_LIBRARY=...): the launcher prints the shader-clock cycles every role of the five-wavefront pair kernel spent per phase
(stderr), this tool the number of slots to divide by and the call's time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

eng = Engine()
key = synthetic.make_key(int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp, n = key.exponent(own), key.n
c = eng.to_device(L.pack(synthetic.random_ciphertexts(key, 1, seed=3), L.limbs_for(key.n_square)))
eng.set_limbs_per_lane(3)
eng.set_wavefronts_per_group(4)
plan = eng.nsquare_plan(n, exp).desc
for _ in range(2):
    t0 = time.perf_counter()
    out = eng.powmod_nsquare_t(c, n, exp)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print(f"products: {plan.n_sqr} squarings + {plan.n_mul} multiplications = {plan.n_sqr + plan.n_mul} slots (+ drains); call {dt * 1e3:.2f} ms (with the probe's synchronisation)")
