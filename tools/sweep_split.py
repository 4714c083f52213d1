#!/usr/bin/env python3
"""ONE powmod_nsquare batch just above the capacity of the wide two-wavefront shape: a single launch (the library's choice
with the split switched off: plain or time-sliced) against the two-launch split the Engine runs on a companion stream
(mx_nsquare_launch_split; forced for every size here) — the data the split's range is fitted to
(profiles/r04_split_launch.txt).   usage: sweep_split.py [key_length ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocols.distributed_keygen_amd import configure_hw_queues

configure_hw_queues(16)
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

eng = Engine()
for key_length in [int(a) for a in sys.argv[1:]] or [2048, 4096]:
    key = synthetic.make_key(key_length, 3, 1)
    own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
    exp, n, n2 = key.exponent(own), key.n, key.n_square
    cap = 8192 if key_length == 2048 else 4096
    sizes = [cap + k * cap // 16 for k in range(0, 13)] + ([10000] if key_length == 2048 else [])
    cts = synthetic.random_ciphertexts(key, max(sizes), seed=7)
    c_all = eng.to_device(L.pack(cts, L.limbs_for(n2)))
    want = None
    print(f"key_length {key_length}: ms for ONE batch (best of 3), single launch vs split; hint = what mx_nsquare_launch_split reports")
    for b in sorted(sizes):
        row = []
        for knob in (1, 2):
            eng.debug_knob("n2_split", knob)
            best = 1e9
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = eng.powmod_nsquare_t(c_all[:b], n, exp)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            row.append(best * 1e3)
            if knob == 1:
                ref = out
            else:
                assert torch.equal(out, ref), "split launch differs from the single launch"
        eng.debug_knob("n2_split", 0)
        hint = eng.nsquare_launch_split(n.bit_length(), b)
        eng.debug_knob("n2_split", 1)
        shape = eng.nsquare_launch_shape(n.bit_length(), b)
        ts = eng.nsquare_launch_timesliced(n.bit_length(), b)
        eng.debug_knob("n2_split", 0)
        print(f"{b:7d}  single {row[0]:7.2f} ({shape[1]}x{shape[4]}w{' ts' if ts[0] else ''})   split {row[1]:7.2f}   -> {b / min(row) / 1e-3 / 1e3:6.1f} k/s best;  hint: {hint}", flush=True)
    got = L.unpack(eng.to_host(out[:2]))
    assert got == [pow(c, exp, n2) for c in cts[:2]]
