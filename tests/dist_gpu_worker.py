#!/usr/bin/env python3
"""One rank of the GPU test of protocols/distributed_keygen_amd/dist.py with the REAL engine (launched as a child
process by tests/test_gpu_dist.py; never imported by pytest).

    dist_gpu_worker.py <rank> <world> <port> <backend>

backend "nccl": RCCL (one rank per GPU; on a one-GPU box only world = 1 is possible — the whole process-group
path, communicator, all_gather_into_tensor on device tensors);  backend "gloo": several ranks share GPU 0 (the
world > 1 code paths — slicing, padding of ragged shards, gather order — on device tensors).
Every sharded_* function is checked against CPython pow() / the oracle on the full batch.  Prints "ok <rank>"."""
import os
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def main() -> None:
    rank, world, port, backend = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from protocols.distributed_keygen_amd import Engine, dist as mxdist, limbs as L, synthetic

    eng = Engine(0)
    ints = lambda t: L.unpack(eng.to_host(t))
    rows = lambda vals, limbs: eng.to_device(L.pack(vals, limbs))
    rng = random.Random(4242)                      # the same (replicated) inputs on every rank
    # ---- partial decryptions modulo N^2, ragged over the ranks (DK:463-466)
    key = synthetic.make_key(1024, 3, 1)
    n, n2 = key.n, key.n_square
    own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
    exp = key.exponent(own)
    cts = synthetic.random_ciphertexts(key, 37, seed=5)
    limbs2 = L.limbs_for(n2)
    got = ints(mxdist.sharded_powmod_nsquare(eng, rows(cts, limbs2), n, exp))
    assert got == [pow(c, exp, n2) for c in cts]
    # ---- recombination with one inconsistent ciphertext: plaintext + status as ONE gathered row (DK:510-515)
    parts = []
    for i in (1, 2, 3):
        e = key.exponent(i)
        parts.append([pow(c, e, n2) if e >= 0 else pow(pow(c, -1, n2), -e, n2) for c in cts])
    parts[2][11] = (parts[2][11] * 3) % n2
    pt = torch.stack([rows(p, limbs2) for p in parts])
    m_t, st = mxdist.sharded_combine(eng, pt, n, key.theta_inv)
    want_st = [1 if k == 11 else 0 for k in range(len(cts))]
    assert st.tolist() == want_st
    for k, m in enumerate(ints(m_t)):
        if k != 11:
            x = parts[0][k] * parts[1][k] * parts[2][k] % n2
            assert m == (x - 1) // n * key.theta_inv % n
    # ---- shared-modulus and per-candidate modexps
    mod = rng.getrandbits(700) | (1 << 699) | 1
    e1 = rng.getrandbits(650)
    bases = [rng.randrange(mod) for _ in range(21)]
    assert ints(mxdist.sharded_powmod_shared(eng, rows(bases, L.limbs_for(mod)), mod, e1)) == [pow(b, e1, mod) for b in bases]
    mods = [rng.getrandbits(515) | (1 << 514) | 1 for _ in range(7)]
    exps = [rng.getrandbits(513) for _ in mods]
    flat = [rng.randrange(m) for m in mods for _ in range(6)]
    got = ints(mxdist.sharded_powmod_multi(eng, rows(flat, 17), mods, exps, 6))
    assert got == [pow(b, exps[k // 6], mods[k // 6]) for k, b in enumerate(flat)]
    # ---- sieve verdict bytes
    primes = oracle.small_prime_list(2000)
    cands = [rng.getrandbits(515) | 1 for _ in range(203)]
    got = mxdist.sharded_sieve(eng, rows(cands, 17), primes)
    assert [bool(x) for x in got.tolist()] == [oracle.small_prime_divisors_test(primes, c) for c in cands]
    # ---- the v-calculation of a keygen round (fused Jacobi -> first-keep selection -> modexps per rank), ragged
    cm = [c for c in cands if not oracle.small_prime_divisors_test(primes, c)][:9]
    ce = [rng.getrandbits(512) for _ in cm]
    gens = [rng.randrange(m) for m in cm for _ in range(24)]
    gens[24:48] = [0] * 24                                      # a candidate without any Jacobi-1 generator
    v_t, cnt = mxdist.sharded_biprime_v(eng, rows(gens, 17), cm, ce, 24, 8)
    want_v, want_c = [], []
    for c, (m, e) in enumerate(zip(cm, ce)):
        kept = [g for g in gens[c * 24 : (c + 1) * 24] if oracle.jacobi_symbol(g, m) == 1][:8]
        want_c.append(len(kept))
        want_v += [pow(g, e, m) for g in kept] + [None] * (8 - len(kept))
    assert cnt.tolist() == want_c and want_c[1] == 0
    assert all(w is None or w == g for w, g in zip(want_v, ints(v_t)))
    # ---- the vote: per-slot pass bytes of all candidates on every rank (DK:1331-1360)
    m0 = cm[0]
    va = [rng.randrange(m0) for _ in range(10)]
    vb = [rng.randrange(m0) for _ in range(10)]
    v1 = [a * b % m0 for a, b in zip(va, vb)]
    v1[3] = (m0 - v1[3]) % m0                                   # -product passes too
    v1[7] = (v1[7] + 1) % m0                                    # a failing slot
    v = torch.stack([rows(v1, 17), rows(va, 17), rows(vb, 17)]).reshape(3, 5, 2, 17)
    votes = mxdist.sharded_biprime_vote(eng, v, [m0] * 5)
    assert votes.tolist() == [[1, 1], [1, 1], [1, 1], [1, 0], [1, 1]]
    # ---- a caller that holds only its shard, incl. fewer rows than ranks
    lo, hi = mxdist.shard_bounds(21, rank, world)
    local = eng.powmod_shared_t(rows(bases[lo:hi], L.limbs_for(mod)), mod, e1) if hi > lo else rows([], L.limbs_for(mod))
    assert ints(mxdist.all_gather_rows(local, 21)) == [pow(b, e1, mod) for b in bases]
    lo, hi = mxdist.shard_bounds(1, rank, world)
    local = eng.powmod_shared_t(rows(bases[lo:hi], L.limbs_for(mod)), mod, e1) if hi > lo else torch.zeros((0, L.limbs_for(mod)), dtype=torch.int32, device=eng.device)
    assert ints(mxdist.all_gather_rows(local, 1)) == [pow(bases[0], e1, mod)]
    # ---- shard-only inputs: every rank packs and uploads ONLY its slice (total= names the whole batch)
    lo, hi = mxdist.shard_bounds(len(cts), rank, world)
    assert ints(mxdist.sharded_powmod_nsquare(eng, rows(cts[lo:hi], limbs2), n, exp, total=len(cts))) == [pow(c, exp, n2) for c in cts]
    m2_t, st2 = mxdist.sharded_combine(eng, torch.stack([rows(p[lo:hi], limbs2) for p in parts]), n, key.theta_inv, total=len(cts))
    assert st2.tolist() == want_st and torch.equal(m2_t[:11], m_t[:11]) and torch.equal(m2_t[12:], m_t[12:])
    lo, hi = mxdist.shard_bounds(len(cm), rank, world)
    v2_t, cnt2 = mxdist.sharded_biprime_v(eng, rows(gens[lo * 24 : hi * 24], 17), cm[lo:hi], ce[lo:hi], 24, 8, total=len(cm))
    assert cnt2.tolist() == want_c and torch.equal(v2_t, v_t)
    lo, hi = mxdist.shard_bounds(5, rank, world)
    assert mxdist.sharded_biprime_vote(eng, v[:, lo:hi].contiguous(), [m0] * (hi - lo), total=5).tolist() == votes.tolist()
    lo, hi = mxdist.shard_bounds(len(cands), rank, world)
    assert torch.equal(mxdist.sharded_sieve(eng, rows(cands[lo:hi], 17), primes, total=len(cands)), got)
    lo, hi = mxdist.shard_bounds(1, rank, world)       # an empty shard on rank 1
    one = mxdist.sharded_powmod_multi(eng, rows(flat[lo * 6 : hi * 6], 17), mods[lo:hi], exps[lo:hi], 6, total=1)
    assert ints(one) == [pow(b, exps[0], mods[0]) for b in flat[:6]]
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print(f"ok {rank} world={world} backend={backend}", flush=True)


if __name__ == "__main__":
    main()
