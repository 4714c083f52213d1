#!/usr/bin/env python3
"""Does the stream -> hardware-queue mapping explain slow chunked batches?  Runs the int-level 40 000
batch (4 chunks on 4 side streams) after other streams have been used, with normal / high priority."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[2] if len(sys.argv) > 2 else "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

mode = sys.argv[1] if len(sys.argv) > 1 else "fresh"
eng = Engine()
key = synthetic.make_key(2048, 3, 1)
exp = next(abs(key.exponent(i)) for i in (1, 2, 3) if key.exponent(i) > 0)
cts = synthetic.random_ciphertexts(key, 40000)
rows = eng.to_device(L.pack(cts[:10000], L.limbs_for(key.n_square)))
if mode != "fresh":
    pre = [torch.cuda.Stream() for _ in range(4)]
    for s in pre:
        with torch.cuda.stream(s):
            eng.powmod_nsquare_t(rows, key.n, exp)
    torch.cuda.synchronize()
if mode == "high":
    eng._side_streams = [torch.cuda.Stream(priority=-1) for _ in range(8)]
if mode == "many":
    eng._side_streams = [torch.cuda.Stream() for _ in range(8)]
for rep in range(4):
    t0 = time.perf_counter()
    out = eng.powmod_nsquare_batch(cts, exp, key.n)
    dt = time.perf_counter() - t0
    print(f"{mode:6s} queues {os.environ['GPU_MAX_HW_QUEUES']}: {dt*1e3:7.1f} ms  {40000/dt/1e3:6.1f} k/s  wait {eng.last_timing['wait_for_gpu_s']*1e3:.1f} ms")
