#!/usr/bin/env python3
"""Randomised differential soak of what round 4 added, against CPython pow / sympy on all host cores (GPU box):
  * powmod_nsquare in random launch shapes (incl. the friendly-modulus one-wavefront instances: moduli on both sides of
    the room threshold), random segment counts, batch sizes with ragged tails, the split launch forced on;
  * the generic kernel in every lane geometry incl. the latency instances (friendly products with N~ + 1 derived on the
    device), per-group moduli of different lengths, exponents 0 / 1 / random;
  * the 257-word Jacobi instance (top limbs in LDS) incl. its safety net.
usage: soak_round4.py [seed] [seconds]"""
import multiprocessing as mp
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocols.distributed_keygen_amd import configure_hw_queues

configure_hw_queues(16)


def _jac(args):
    from sympy import jacobi_symbol

    return int(jacobi_symbol(*args))


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
    from protocols.distributed_keygen_amd import Engine

    eng = Engine()
    rng = random.Random(seed)
    pool = mp.Pool(16)
    t0 = time.time()
    done = {"nsquare": 0, "generic": 0, "jacobi257": 0}
    rounds = 0
    while time.time() - t0 < budget:
        rounds += 1
        # ---- pair kernel
        nb = rng.choice([2050, 2051, 2053, 2054, 2056, 2058, 2075, 3075, 4099, 4102, 4140, 4150, 1028, 515, 131])
        n = rng.getrandbits(nb) | (1 << (nb - 1)) | 1
        n2 = n * n
        e = rng.getrandbits(rng.choice([2 * nb + 90, nb, 64, 17]))
        lpl, wpg = rng.choice([(18, 1), (18, 1), (18, 1), (9, 1), (18, 2), (9, 2), (3, 2), (0, 0)])
        if lpl == 3 and nb > 4140:
            lpl, wpg = 18, 1
        batch = rng.choice([1, 5, 16, 17, 63, 100, 257])
        bases = [rng.randrange(n2) for _ in range(batch)] + [n * rng.randrange(n), 0, 1, n2 - 1]
        eng.set_limbs_per_lane(lpl)
        eng.set_wavefronts_per_group(wpg)
        eng.set_segments(rng.choice([0, 1, 2, 3, 5, 9]))
        eng.debug_knob("n2_friendly_1w", rng.choice([0, 0, 0, 1]))
        got = eng.powmod_nsquare_batch(bases, e, n)
        want = pool.starmap(pow, [(b, e, n2) for b in bases], chunksize=4)
        assert got == want, ("nsquare", rounds, nb, e.bit_length(), lpl, wpg)
        done["nsquare"] += len(bases)
        eng.set_segments(0)
        eng.debug_knob("n2_friendly_1w", 0)
        # ---- generic kernel, per-group moduli
        mb = rng.choice([40, 52, 53, 139, 140, 313, 600, 661, 662, 1028, 1357, 1358, 2053, 2749, 2750, 4100, 5533])
        glpl = rng.choice([0, 3, 3, 9, 18])
        groups = rng.choice([1, 2, 3, 7])
        gsize = rng.choice([1, 3, 40, 41])
        mods = [rng.getrandbits(mb - rng.randrange(0, 3)) | (1 << (mb - 4)) | 1 for _ in range(groups)]
        exps = [rng.choice([0, 1, 2, rng.getrandbits(rng.choice([mb, mb // 2, 70])) ]) for _ in mods]
        rows = [[rng.randrange(m) for _ in range(gsize)] for m in mods]
        rows[0][0] = mods[0] - 1
        eng.set_limbs_per_lane(glpl)
        eng.set_wavefronts_per_group(0)
        if groups == 1:
            got = [eng.powmod_batch(rows[0], exps[0], mods[0])]
        else:
            got = eng.powmod_batch_multi(rows, exps, mods)
        want = pool.starmap(pow, [(b, ex, m) for r, ex, m in zip(rows, exps, mods) for b in r], chunksize=8)
        assert [x for r in got for x in r] == want, ("generic", rounds, mb, glpl, groups, gsize)
        done["generic"] += len(want)
        eng.set_limbs_per_lane(0)
        # ---- Jacobi, 257 words (every fourth round: sympy at 8200 bits is slow)
        if rounds % 4 == 0:
            jb = rng.choice([8197, 8200, 8224, 6000, 4200])
            jm = [rng.getrandbits(jb) | (1 << (jb - 1)) | 1, 3 * 5 * 7 * (rng.getrandbits(jb - 8) | 1)]
            jv = [[rng.choice([rng.randrange(m), rng.getrandbits(40), m - rng.getrandbits(30), 21 * rng.getrandbits(jb - 10)]) % m for _ in range(24)] for m in jm]
            cut = rng.choice([0, 0, 3, 40])
            eng.debug_knob("jacobi_max_batches", cut)
            gotj = eng.jacobi_batch(jv, jm)
            eng.debug_knob("jacobi_max_batches", 0)
            wantj = pool.map(_jac, [(v, m) for r, m in zip(jv, jm) for v in r], chunksize=3)
            assert [x for r in gotj for x in r] == wantj, ("jacobi", rounds, jb, cut)
            done["jacobi257"] += len(wantj)
    pool.close()
    print(f"soak ok (seed {seed}): {rounds} rounds in {time.time() - t0:.0f} s — {done['nsquare']} pair-kernel modexps over random shapes / segments / "
          f"friendly on-off, {done['generic']} generic modexps over lane geometries 0/3/9/18, {done['jacobi257']} Jacobi symbols on the 257-word instance, all bit-exact")


if __name__ == "__main__":
    main()
