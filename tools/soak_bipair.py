#!/usr/bin/env python3
"""Randomised differential soak of the five-wavefront latency form of the pair kernel (csrc/mx_bipair.hpp) against CPython pow on
the host cores: modulus lengths over the whole range the form takes (groups of 16 and 32 lanes, lengths around the geometry
steps), random and special moduli, bases 0 / 1 / multiples of N / N^2 - 1 / random, exponents from one bit to full length with
sliding and fixed-window tapes, batches from 1 to several workgroups per compute unit, now and then two launches at once on two streams.
usage: soak_bipair.py [seed] [seconds]
tests/test_gpu_soak_slices.py runs soak(engine, seed, rounds_limit=...) for fixed seeds inside `pytest -m gpu`."""
import multiprocessing as mp
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def soak(eng, seed, seconds=None, rounds_limit=None, procs=16):
    """Runs until `seconds` have passed or `rounds_limit` rounds are done; returns (rounds, modexps compared)."""
    import ctypes

    import torch

    from protocols.distributed_keygen_amd import limbs as L

    rng = random.Random(seed)
    pool = mp.Pool(procs)
    side = torch.cuda.Stream()
    t0 = time.time()
    rounds = done = 0
    # modulus lengths with an instance of the form (probed through the library's own query)
    lens = []
    for nb in list(range(600, 2800, 13)) + list(range(2800, 5700, 61)) + [1026, 1027, 1028, 1029, 2050, 2051, 2052, 2053, 4098, 4099, 4100, 4102]:
        k, l, wv, fr, ts = (ctypes.c_int() for _ in range(5))
        if eng.lib.mx_nsquare_launch_instance(nb, 1, 3, 4, k, l, wv, fr, ts) == 0:
            lens.append(nb)
    assert lens and min(lens) < 900 and max(lens) > 5000, (min(lens), max(lens))
    try:
        while (seconds is None or time.time() - t0 < seconds) and (rounds_limit is None or rounds < rounds_limit):
            rounds += 1
            nb = rng.choice(lens)
            n = rng.choice([rng.getrandbits(nb) | (1 << (nb - 1)) | 1] * 3 + [(1 << nb) - 1, (1 << (nb - 1)) + 1, ((1 << nb) - 1) ^ (1 << (nb // 2))]) | 1
            n2 = n * n
            ebits = rng.choice([1, 2, 3, 17, 64, 200, nb, 2 * nb + 90]) if rng.random() < 0.7 else rng.randint(1, 2 * nb + 90)
            e = rng.choice([rng.getrandbits(ebits) | (1 << (ebits - 1)), (1 << ebits) - 1, 1 << (ebits - 1)])
            batch = rng.choice([1, 1, 2, 3, 5, 17, 64, 130, 513, 700]) if ebits < 400 else rng.choice([1, 2, 3, 9])
            bases = ([rng.choice([0, 1, n, n2 - 1, n + 1, n * rng.randrange(n)]) for _ in range(min(batch, 4))] + [rng.randrange(n2) for _ in range(batch)])[:batch]
            eng.set_limbs_per_lane(3)
            eng.set_wavefronts_per_group(4)
            eng.set_fixed_window(rng.random() < 0.3)
            want = pool.starmap(pow, [(b, e, n2) for b in bases], chunksize=4)
            if rng.random() < 0.25 and batch > 1:
                c = eng.to_device(L.pack(bases, L.limbs_for(n2)))
                outs = [eng.powmod_nsquare_t(c, n, e)]
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    outs.append(eng.powmod_nsquare_t(c, n, e))
                torch.cuda.synchronize()
                for o in outs:
                    assert L.unpack(eng.to_host(o)) == want, ("two streams", rounds, nb, ebits, batch)
                done += 2 * batch
            else:
                got = eng.powmod_nsquare_batch(bases, e, n)
                assert got == want, ("four wavefronts", rounds, nb, ebits, batch, [i for i, (x, y) in enumerate(zip(got, want)) if x != y][:5])
                done += batch
    finally:
        eng.set_fixed_window(False)
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)
        pool.terminate()
    return rounds, done


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
    from protocols.distributed_keygen_amd import Engine, configure_hw_queues

    configure_hw_queues(16)
    t0 = time.time()
    rounds, done = soak(Engine(), seed, seconds=budget)
    print(f"soak_bipair seed {seed}: {rounds} rounds, {done} modexps in {time.time() - t0:.0f} s, all bit-exact against pow")


if __name__ == "__main__":
    main()
