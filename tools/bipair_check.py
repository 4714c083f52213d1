#!/usr/bin/env python3
"""The five-wavefront latency form of the pair kernel (csrc/mx_bipair.hpp) against CPython pow: random and special moduli
of key_length 1024 / 2048 size, exponents from a few bits to full length, batches of 1 .. 600, and its time for ONE
ciphertext beside the two-wavefront form's.   usage: bipair_check.py [seed]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

eng = Engine()
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
for nb in (2051, 2053, 1027, 1029, 2050, 1500, 2600, 4099, 4102, 5300, 5700):
    for trial in range(3):
        n = [rng.getrandbits(nb) | (1 << (nb - 1)) | 1, (1 << nb) - 1, (1 << (nb - 1)) + 1][trial]
        n2 = n * n
        for ebits in (1, 2, 17, 200, nb, 2 * nb + 90):
            e = rng.getrandbits(ebits) | (1 << (ebits - 1))
            batch = rng.choice([1, 2, 3, 5, 64, 130])
            bases = ([0, 1, n, n2 - 1, n + 1] + [rng.randrange(n2) for _ in range(batch)])[:max(batch, 1)]
            eng.set_limbs_per_lane(3)
            eng.set_wavefronts_per_group(4)
            try:
                got = eng.powmod_nsquare_batch(bases, e, n)
            except Exception as exc:
                if nb in (2600, 5700) and "MX_ERR_SIZE" in str(exc):        # between / beyond the form's instances: refused, not computed
                    break
                print("ERROR", nb, trial, ebits, batch, type(exc).__name__, exc)
                bad += 1
                break
            want = [pow(b, e, n2) for b in bases]
            if got != want:
                wrong = [i for i, (x, y) in enumerate(zip(got, want)) if x != y]
                print("MISMATCH", nb, trial, ebits, batch, wrong[:5], hex(got[wrong[0]])[:40], hex(want[wrong[0]])[:40])
                bad += 1
print("mismatching configurations:", bad)
# timing of one ciphertext, key_length 2048 full exponent
key = synthetic.make_key(2048, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp, n = key.exponent(own), key.n
for batch in (1, 64, 256, 512, 1024):
    c = eng.to_device(L.pack(synthetic.random_ciphertexts(key, batch, seed=3), L.limbs_for(key.n_square)))
    res = {}
    for wpg in (2, 4):
        eng.set_limbs_per_lane(3)
        eng.set_wavefronts_per_group(wpg)
        out = eng.powmod_nsquare_t(c, n, exp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = eng.powmod_nsquare_t(c, n, exp)
        torch.cuda.synchronize()
        res[wpg] = ((time.perf_counter() - t0) / 5 * 1e3, out.clone())
    same = bool((res[2][1] == res[4][1]).all())
    print(f"batch {batch}: two wavefronts {res[2][0]:.2f} ms, five-wavefront form {res[4][0]:.2f} ms, identical {same}")
sys.exit(1 if bad else 0)
