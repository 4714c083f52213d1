#!/usr/bin/env python3
"""Mismatch census of powmod_nsquare launches against CPython pow over shapes and segment counts (developer probe)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
import torch
from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic
eng = Engine()
key = synthetic.make_key(2048, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp, n, n2 = key.exponent(own), key.n, key.n_square
B = int(os.environ.get("FR_B", "2048"))
cts = synthetic.random_ciphertexts(key, B, seed=11)
with mp.Pool(16) as pool:
    want = pool.starmap(pow, [(c, exp, n2) for c in cts], chunksize=16)
c_all = eng.to_device(L.pack(cts, L.limbs_for(n2)))
for lpl, wpg in ((18, 1), (9, 1), (3, 2), (9, 2)):
    for seg in (1, 4):
        eng.set_limbs_per_lane(lpl); eng.set_wavefronts_per_group(wpg)
        got = L.unpack(eng.to_host(eng.powmod_nsquare_t(c_all, n, exp, segments=seg)))
        bad = [i for i in range(B) if got[i] != want[i]]
        print(f"L{lpl}x{wpg}w segments {seg}: {len(bad)} mismatches of {B}", bad[:8], flush=True)
        if bad:
            i = bad[0]
            d = (got[i] - want[i]) % n2
            print("   diff mod N:", d % n, " diff // N:", d // n if d % n == 0 else None, " got < n2:", got[i] < n2)

# the same launches from four streams at once (each its own output and workspace), default segments
for lpl, wpg in ((18, 1), (9, 2), (18, 2), (3, 2), (9, 1)):
    eng.set_limbs_per_lane(lpl); eng.set_wavefronts_per_group(wpg)
    print(f"L{lpl}x{wpg}w, time-sliced: {eng.nsquare_launch_timesliced(n.bit_length(), B)}")
    streams = [torch.cuda.Stream() for _ in range(4)]
    for rep in range(2):
        outs = []
        for st in streams:
            with torch.cuda.stream(st):
                outs.append(eng.powmod_nsquare_t(c_all, n, exp))
        torch.cuda.synchronize()
        for k, o in enumerate(outs):
            got = L.unpack(eng.to_host(o))
            bad = [i for i in range(B) if got[i] != want[i]]
            print(f"4 streams rep {rep} stream {k}: {len(bad)} mismatches", bad[:6], flush=True)
