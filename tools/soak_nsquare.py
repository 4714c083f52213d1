#!/usr/bin/env python3
"""Soak test of mx_powmod_nsquare against CPython pow on all host cores (GPU box).

usage: soak_nsquare.py [seed] [trials] [limbs_per_lane 0|9|18]"""
import multiprocessing as mp, random, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    from protocols.distributed_keygen_amd import Engine
    eng = Engine()
    eng.set_limbs_per_lane(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    t0 = time.time(); checked = 0
    with mp.Pool() as pool:
        for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
            nb = rng.choice([2050, 2051, 2052, 2053, 1028, 515, 4099])
            n = rng.getrandbits(nb) | (1 << (nb - 1)) | 1
            e = rng.getrandbits(rng.choice([2 * nb + 90, nb, 64]))
            n2 = n * n
            bases = [rng.randrange(n2) for _ in range(96)] + [n * rng.randrange(n) for _ in range(4)]
            got = eng.powmod_nsquare_batch(bases, e, n)
            want = pool.starmap(pow, [(b, e, n2) for b in bases], chunksize=2)
            assert got == want, (trial, nb, e.bit_length())
            got2 = eng.powmod_batch(bases, e, n2)
            assert got2 == want, ("generic", trial)
            checked += len(bases)
    print(f"soak ok: {checked} modexps bit-exact in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()
