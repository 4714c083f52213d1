#!/usr/bin/env python3
"""Randomised differential soak of what round 5 added, against CPython pow on all host cores (GPU box):
  * the bipartite latency form of the generic kernel (limbs_per_lane 6) over random modulus lengths (every group width 4..64,
    lengths around the geometry steps), special and random moduli, per-group moduli of different — now and then VERY different — lengths, ragged groups,
    exponents 0 / 1 / all-ones / random, every pivot (developer knob) and more lanes per element than needed;
  * partial decryptions with the fixed-window tape in random launch shapes and segment counts.
usage: soak_round5.py [seed] [seconds]
tests/test_gpu_soak_slices.py runs soak(engine, seed, rounds=...) for a few fixed seeds inside `pytest -m gpu`."""
import multiprocessing as mp
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def soak(eng, seed, seconds=None, rounds_limit=None, procs=16):
    """Runs until `seconds` have passed or `rounds_limit` rounds are done (whichever is given); returns (rounds, counts).
    Raises AssertionError on the first row that differs from CPython pow."""
    rng = random.Random(seed)
    pool = mp.Pool(procs)
    t0 = time.time()
    done = {"bipartite": 0, "fixed_window": 0}
    rounds = 0
    try:
        while (seconds is None or time.time() - t0 < seconds) and (rounds_limit is None or rounds < rounds_limit):
            rounds += 1
            # ---- bipartite form
            bits = rng.choice([rng.randint(3, 140), rng.randint(140, 700), rng.choice([1026, 1027, 1028, 1029, 1183, 1184, 2050, 2051, 2052, 2053]),
                               rng.randint(700, 2600), rng.randint(2600, 5359), rng.choice([52, 53, 139, 140, 313, 314, 2574, 2575, 5359])])
            groups = rng.choice([1, 2, 3, 7])
            special = lambda b: rng.choice([(1 << b) - 1, (1 << (b - 1)) + 1, ((1 << b) - 1) ^ (1 << (b // 2)), (1 << b) - (1 << (b // 3)) - 1]) | 1
            wild = rng.random() < 0.2          # a launch has ONE geometry, from its longest modulus: groups with much shorter ones
            glen = [max(2, bits - g) if not (wild and g) else rng.randint(2, bits) for g in range(groups)]
            mods = [max(3, special(b) if rng.random() < 0.25 else rng.getrandbits(b) | (1 << (b - 1)) | 1) for b in glen]
            ebits = rng.choice([1, 2, 17, 64, 150, min(bits, 400)])
            exps = [rng.choice([0, 1, (1 << ebits) - 1, rng.getrandbits(ebits)]) for _ in mods]
            gsize = rng.choice([1, 3, 5, 16, 40]) if bits < 2600 else rng.choice([1, 3])
            rows = [[rng.choice([0, 1, m - 1, rng.randrange(m)]) for _ in range(rng.randint(1, gsize))] for m in mods]
            eng.set_limbs_per_lane(6)
            steps = 3 * (-(-(bits + 35) // 87)) + 3
            pivot_knob = rng.choice([0, 0, 3 * rng.randint(1, max(1, steps // 3 - 1))])
            lanes_knob = rng.choice([0, 0, 0, 16, 64])
            eng.debug_knob("bi_pivot", pivot_knob)
            eng.debug_knob("lat_lanes", lanes_knob)
            assert eng.generic_launch_form(bits, sum(len(r) for r in rows) or 1, groups)[0] == 2
            got = eng.powmod_batch_multi(rows, exps, mods)
            want = pool.starmap(pow, [(b, e, m) for r, e, m in zip(rows, exps, mods) for b in r], chunksize=4)
            flat = [x for r in got for x in r]
            if flat != want:
                bad = [i for i, (x, y) in enumerate(zip(flat, want)) if x != y]
                print("MISMATCH bipartite", {"round": rounds, "bits": bits, "groups": groups, "ebits": ebits, "pivot_knob": pivot_knob, "lanes_knob": lanes_knob,
                                             "mods": [hex(m) for m in mods], "exps": [hex(e) for e in exps], "rows": [[hex(b) for b in r] for r in rows],
                                             "bad": bad[:10], "got": [hex(flat[i]) for i in bad[:4]], "want": [hex(want[i]) for i in bad[:4]]}, flush=True)
            assert flat == want, ("bipartite", rounds, bits, groups, ebits)
            done["bipartite"] += len(flat)
            eng.debug_knob("bi_pivot", 0)
            eng.debug_knob("lat_lanes", 0)
            # ---- fixed-window tape of the pair kernel
            nb = rng.choice([2051, 2053, 2075, 1028, 515, 131, 3075, 4099])
            n = rng.getrandbits(nb) | (1 << (nb - 1)) | 1
            n2 = n * n
            e = rng.getrandbits(rng.choice([2 * nb + 90, nb, 64, 17, 1]))
            lpl, wpg = rng.choice([(18, 1), (9, 1), (18, 2), (9, 2), (3, 2), (0, 0)])
            bases = [rng.randrange(n2) for _ in range(rng.choice([1, 5, 17, 100]))] + [n * rng.randrange(n), 0, 1, n2 - 1]
            eng.set_limbs_per_lane(lpl)
            eng.set_wavefronts_per_group(wpg)
            eng.set_segments(rng.choice([0, 1, 2, 5]))
            eng.set_fixed_window(True)
            got = eng.powmod_nsquare_batch(bases, e, n)
            want = pool.starmap(pow, [(b, e, n2) for b in bases], chunksize=4)
            assert got == want, ("fixed_window", rounds, nb, e.bit_length(), lpl, wpg)
            done["fixed_window"] += len(bases)
            eng.set_fixed_window(False)
            eng.set_segments(0)
            eng.set_wavefronts_per_group(0)
    finally:
        eng.set_fixed_window(False)
        eng.set_segments(0)
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)
        eng.debug_knob("bi_pivot", 0)
        eng.debug_knob("lat_lanes", 0)
        pool.terminate()
    return rounds, done


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
    from protocols.distributed_keygen_amd import Engine, configure_hw_queues

    configure_hw_queues(16)
    t0 = time.time()
    rounds, done = soak(Engine(), seed, seconds=budget)
    print(f"seed {seed}: {rounds} rounds in {time.time() - t0:.0f} s, all bit-exact: {done}")


if __name__ == "__main__":
    main()
