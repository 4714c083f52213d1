#!/usr/bin/env python3
"""Where the milliseconds of ONE int-level call of 10 000 ciphertexts go (Engine.powmod_nsquare_batch, key_length 2048):
wall clock of the call, its stages (Engine.last_timing), and the modexp kernel's own duration (HIP events around the
launch, mx_profile) — back to back and with a host-only pause in front of every call, as a caller's own work would be.
usage: lone_call_probe.py [batch] [pause_ms ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protocols.distributed_keygen_amd import Engine, configure_hw_queues, synthetic

configure_hw_queues(16)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
pauses = [float(v) for v in sys.argv[2:]] or [0.0, 3.0, 30.0]
eng = Engine()
key = synthetic.make_key(2048, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp, n = key.exponent(own), key.n
cts = synthetic.random_ciphertexts(key, batch, seed=7)
eng.powmod_nsquare_batch(cts, exp, n)
eng.profile(True)
for pause in pauses:
    rows = []
    for rep in range(5):
        if pause:
            t_end = time.perf_counter() + pause / 1e3
            while time.perf_counter() < t_end:          # host-only work of the caller
                pass
        torch.cuda.synchronize()
        eng.profile_collect()
        t0 = time.perf_counter()
        eng.powmod_nsquare_batch(cts, exp, n)
        wall = (time.perf_counter() - t0) * 1e3
        kernel_ms, launches = eng.profile_collect()
        tm = eng.last_timing
        rows.append((wall, kernel_ms, tm["pack_s"] * 1e3, tm["copies_and_gpu_s"] * 1e3, tm["unpack_s"] * 1e3))
    rows.sort()
    w, k, p, c, u = rows[len(rows) // 2]
    print(f"batch {batch}, {pause:4.0f} ms of host work in front of every call: wall {w:.2f} ms = pack {p:.2f} + copies and GPU {c:.2f} (modexp kernel {k:.2f}) "
          f"+ unpack {u:.2f} + {w - p - c - u:.2f}; all calls: " + " ".join(f"{r[0]:.1f}/{r[1]:.1f}" for r in rows), flush=True)
