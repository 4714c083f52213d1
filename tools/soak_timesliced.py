#!/usr/bin/env python3
"""Randomised differential soak of the time-sliced form of the two-wavefront pair kernel (round 5's unit scheduler and
hand-over): random key lengths (every group width that has a time-sliced instance, at 9 and 18 limbs per lane, friendly
and plain moduli), batches from a few groups to 1.6 x the resident pairs (ragged last groups), 1 .. 16 units per group,
1 .. 3 workgroups per CU, exponents from a few bits to full length, one launch alone or two on two streams at once —
every row compared with the plain one-wavefront launch of the same input, and a sample of rows with CPython pow.
usage: soak_timesliced.py [seed] [seconds]
tests/test_gpu_soak_slices.py runs soak(engine, seed, launches_limit=...) for fixed seeds inside `pytest -m gpu`."""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def soak(eng, seed, seconds=None, launches_limit=None, max_rows=None):
    """Runs until `seconds` have passed or `launches_limit` launches were compared; returns (launches, rows, by_shape).
    max_rows bounds the size of one launch (the bounded slice inside the GPU suite).  AssertionError on the first
    launch with a row that differs from the plain one-wavefront launch (itself sampled against CPython pow)."""
    import torch

    from protocols.distributed_keygen_amd import limbs as L

    budget = seconds
    rng = random.Random(seed)
    side = torch.cuda.Stream()
    t0 = time.time()
    launches = rows_checked = 0
    by_shape = {}
    try:
        while (budget is None or time.time() - t0 < budget) and (launches_limit is None or launches < launches_limit):
            nb = rng.choice([131, 300, 515, 1027, 1030, 2051, 2053, 2075, 3075, 4099, 4160])
            lpl = rng.choice([9, 18])
            n = rng.getrandbits(nb) | (1 << (nb - 1)) | 1
            n2 = n * n
            limbs2 = L.limbs_for(n2)
            # elements per group of this geometry, resident pairs of one workgroup per CU
            eng.set_limbs_per_lane(lpl); eng.set_wavefronts_per_group(2)
            k = eng.nsquare_launch_shape(nb, 64)[0]
            gpw = 64 // k
            pairs = 512
            scale = rng.choice([0.02, 0.3, 0.9, 1.0, 1.1, 1.22, 1.6]) if nb >= 1027 else rng.choice([0.02, 0.2, 1.05])
            batch = max(1, int(pairs * gpw * scale) + rng.randint(-gpw, gpw))
            if nb >= 3075:
                batch = min(batch, 3000)
            if max_rows:
                batch = min(batch, max_rows)
            ebits = rng.choice([40, 200, nb]) if batch * nb > 3_000_000 else rng.choice([17, 200, nb, 2 * nb + 90])
            e = rng.getrandbits(ebits) | (1 << (ebits - 1)) | 1
            bases = [0, 1, n, n2 - 1][:batch] + [rng.randrange(n2) for _ in range(max(0, batch - 4))]
            c = eng.to_device(L.pack(bases, limbs2))
            eng.set_limbs_per_lane(18 if nb >= 300 else 9); eng.set_wavefronts_per_group(1); eng.debug_knob("n2_timeslice", 1)
            want = eng.powmod_nsquare_t(c, n, e, segments=1).clone()
            torch.cuda.synchronize()
            idx = sorted(set([0, 1, batch - 1] + [rng.randrange(batch) for _ in range(5)]))
            assert L.unpack(eng.to_host(want[idx])) == [pow(bases[i], e, n2) for i in idx], ("plain", nb, batch)
            r = rng.choice([1, 2, 3]) if lpl == 9 else 1
            units = rng.choice([1, 2, 3, 5, 8, 12, 16])
            eng.set_limbs_per_lane(lpl); eng.set_wavefronts_per_group(2); eng.debug_knob("n2_timeslice", 16 + r)
            sliced = eng.nsquare_launch_timesliced(nb, batch)[0] > 0
            two = rng.random() < 0.3
            outs = [eng.powmod_nsquare_t(c, n, e, segments=units)]
            if two:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    outs.append(eng.powmod_nsquare_t(c, n, e, segments=units))
            torch.cuda.synchronize()
            for o in outs:
                bad = int((o != want).any(dim=1).sum())
                assert bad == 0, ("time-sliced", nb, lpl, k, batch, r, units, ebits, two, bad)
            launches += len(outs)
            rows_checked += batch * len(outs)
            key = (lpl, k, "sliced" if sliced else "plain")
            by_shape[key] = by_shape.get(key, 0) + len(outs)
            eng.debug_knob("n2_timeslice", 0)
    finally:
        eng.debug_knob("n2_timeslice", 0); eng.set_limbs_per_lane(0); eng.set_wavefronts_per_group(0)
    return launches, rows_checked, by_shape


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
    from protocols.distributed_keygen_amd import Engine, configure_hw_queues

    configure_hw_queues(16)
    t0 = time.time()
    launches, rows_checked, by_shape = soak(Engine(), seed, seconds=budget)
    print(f"soak_timesliced seed {seed}: {launches} launches, {rows_checked} rows bit-identical to the plain launch in {time.time() - t0:.0f} s; "
          "launches per (limbs per lane, lanes per element, form): " + ", ".join(f"{k}: {v}" for k, v in sorted(by_shape.items())))


if __name__ == "__main__":
    main()
