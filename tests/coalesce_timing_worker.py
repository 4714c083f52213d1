#!/usr/bin/env python3
"""Wall-clock side of the coalescing tests, in a process of its own (child of tests/test_gpu_standin.py; never imported by
pytest): a process that has used dozens of HIP streams — the GPU suite's has — is time-sliced by the queue scheduler and
runs everything 1.5-2x slower, so times measured inside the suite say nothing about a user's process.
Prints two JSON lines:
  {"decrypt_bursts_ms": [...]}                       3 parties x 256 concurrent decrypt() at key_length 2048
  {"keygen_busy_ms_per_round": [separate, shared]}   3 co-located parties, key_length 1024, 1024 candidates per round"""
import json
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main() -> None:
    import standin_harness as sh
    from protocols.distributed_keygen_amd import Engine, patch, synthetic

    patch.install(engine=None, package=sh.PACKAGE)          # before the first GPU call: install() asks for 16 hardware queues
    patch.uninstall()
    eng = Engine()
    key = synthetic.make_key(2048, 3, 1)
    rng = random.Random(2048)
    count = 256
    msgs = [rng.randrange(key.n) for _ in range(count)]
    cts, seen = [], set()
    while len(cts) < count:
        c = synthetic.encrypt(key, msgs[len(cts)], rng)
        tag = bin(c).zfill(32)[2:34]
        if tag not in seen:
            seen.add(tag)
            cts.append(c)
    patch.install(engine=eng, package=sh.PACKAGE)
    try:
        parties = sh.parties_for_key(key)
        got = sh.decrypt_many(parties, sh.ciphertexts(key, cts))          # prepares the per-key plans
        assert [e.value for e in got] == msgs * 3
        walls = []
        for _ in range(4):
            cobjs = sh.ciphertexts(key, cts)
            t0 = time.perf_counter()
            got = sh.decrypt_many(parties, cobjs)
            walls.append(time.perf_counter() - t0)
            assert [e.value for e in got] == msgs * 3
        print(json.dumps({"decrypt_bursts_ms": [round(w * 1e3, 2) for w in walls]}), flush=True)
    finally:
        patch.uninstall()
    busy = []
    for merge in (False, True):
        patch.install(engine=eng, package=sh.PACKAGE)
        try:
            rc = patch.round_coalescer(sh.PACKAGE)
            rc.merge = merge
            got = sh.keygen(seed=12, key_length=1024, batch_size=1024, prime_threshold=2000, correct_param=40)
            st = dict(rc.stats)
        finally:
            patch.uninstall()
        assert len(set(got)) == 1
        busy.append(round(st["busy_s"] / (st["sieve_requests"] // 3) * 1e3, 2))
    print(json.dumps({"keygen_busy_ms_per_round": busy}), flush=True)


if __name__ == "__main__":
    main()
