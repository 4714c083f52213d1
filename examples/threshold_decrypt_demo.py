#!/usr/bin/env python3
"""Three parties of a threshold-Paillier key decrypt a batch of ciphertexts on one GPU.

Mirrors the reference's local example (README.md:224-276 of TNO-MPC/protocols.distributed_keygen)
at the level of the hot path: the key is synthetic (protocols.distributed_keygen_amd.synthetic —
the MPC key generation itself is the reference's control plane and not part of this repository),
the message exchange between parties is a Python dict, every big-integer operation runs through the
C ABI.   python examples/threshold_decrypt_demo.py [--key-length 2048] [--count 1000]
"""
import argparse
import random
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--key-length", type=int, default=2048)
    ap.add_argument("--count", type=int, default=1000)
    args = ap.parse_args()
    from protocols.distributed_keygen_amd import Engine, synthetic
    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, PlainCiphertext, ShareView

    eng = Engine()
    key = synthetic.make_key(args.key_length, 3, 1)
    parties = {
        i: GpuPaillierSharedKey(key.n, key.t, i, ShareView({i: key.shares[i]}, key.degree, key.n_fac), key.theta, engine=eng)
        for i in (1, 2, 3)
    }
    rng = random.Random(1)
    messages = [rng.randrange(key.n) for _ in range(args.count)]
    t0 = time.perf_counter()
    ciphertexts = eng.encrypt_batch(messages, [rng.randrange(1, key.n) for _ in messages], key.n)
    t1 = time.perf_counter()
    # every party: partial decryptions of the whole sequence (distributed_keygen.py:463-466)
    partials = {i: k.partial_decrypt_batch([PlainCiphertext(c, key.n) for c in ciphertexts]) for i, k in parties.items()}
    t2 = time.perf_counter()
    # party 1 recombines what it "received" (distributed_keygen.py:494-515)
    plain = parties[1].decrypt_batch([{i: partials[i][e] for i in parties} for e in range(args.count)])
    t3 = time.perf_counter()
    assert plain == messages
    print(f"key_length {args.key_length}: {args.count} ciphertexts  encrypt {t1 - t0:.3f}s  "
          f"3 x partial-decrypt {t2 - t1:.3f}s  recombine {t3 - t2:.3f}s  (host packing included) — round trip OK")


if __name__ == "__main__":
    main()
