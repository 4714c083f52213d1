#!/usr/bin/env python3
"""A few lone powmod_nsquare launches of chosen shapes, for counter passes (rocprofv3 --pmc SQC_ICACHE_* ...):
usage: icache_probe.py key_length batch limbs_per_lane wavefronts_per_group [...more quadruples]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic
eng = Engine()
args = [int(a) for a in sys.argv[1:]]
for key_length, b, lpl, wpg in zip(args[0::4], args[1::4], args[2::4], args[3::4]):
    key = synthetic.make_key(key_length, 3, 1)
    own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
    cts = synthetic.random_ciphertexts(key, b, seed=7)
    c = eng.to_device(L.pack(cts, L.limbs_for(key.n_square)))
    eng.set_limbs_per_lane(lpl); eng.set_wavefronts_per_group(wpg); eng.debug_knob("n2_timeslice", 1)
    for _ in range(3):
        eng.powmod_nsquare_t(c, key.n, key.exponent(own), segments=1)
        torch.cuda.synchronize()
