#!/usr/bin/env python3
"""Where does a one-shot int-level batch spend its time?  Repeats Engine.powmod_nsquare_batch on 40 000
ciphertexts back to back and after an idle pause (clock ramp), printing the engine's own split."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protocols.distributed_keygen_amd import Engine, synthetic

eng = Engine()
key = synthetic.make_key(2048, 3, 1)
exp = next(abs(key.exponent(i)) for i in (1, 2, 3) if key.exponent(i) > 0)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
cts = synthetic.random_ciphertexts(key, total)
for label, pause in (("warm-up", 0), ("back to back", 0), ("back to back", 0), ("after 0.5 s idle", 0.5), ("back to back", 0), ("after 2 s idle", 2.0)):
    time.sleep(pause)
    t0 = time.perf_counter()
    out = eng.powmod_nsquare_batch(cts, exp, key.n)
    dt = time.perf_counter() - t0
    print(f"{label:18s} {dt*1e3:7.1f} ms  {total/dt/1e3:6.1f} k/s  {eng.last_timing}")
assert out[7] == pow(cts[7], exp, key.n_square)
