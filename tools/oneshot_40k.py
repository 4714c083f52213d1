#!/usr/bin/env python3
"""One-shot long sequence: how fast can 40 000 ciphertexts go through when they arrive at once?
Tensor-level (rows already on the device): chunk count x geometry x segments."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

eng = Engine()
key = synthetic.make_key(2048, 3, 1)
n = key.n
exp = abs(key.exponent(2)) if key.exponent(2) > 0 else abs(key.exponent(1))
total = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
cts = synthetic.random_ciphertexts(key, total)
rows = eng.to_device(L.pack(cts, L.limbs_for(n * n)))
streams = [torch.cuda.Stream() for _ in range(16)]
ref = None
for lpl, seg, chunks in ((18, 1, 4), (18, 4, 4), (18, 8, 4), (18, 4, 8), (18, 8, 8), (9, 4, 4), (9, 4, 8), (0, 4, 5), (18, 16, 4), (18, 4, 1), (18, 8, 1), (18, 16, 1), (18, 32, 1)):
    eng.set_limbs_per_lane(lpl); eng.set_segments(seg)
    per = -(-total // chunks)
    best = 1e9
    for rep in range(3):
        outs = []
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(chunks):
            with torch.cuda.stream(streams[k]):
                outs.append(eng.powmod_nsquare_t(rows[k * per:(k + 1) * per], n, exp))
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    full = torch.cat(outs)
    if ref is None: ref = full
    assert torch.equal(full, ref)
    print(f"lpl {lpl:2d} segments {seg:2d} chunks {chunks}: {best*1e3:7.1f} ms  {total/best/1e3:6.1f} k/s")
