"""world_size-2 gloo tests of the sharding layer (protocols/distributed_keygen_amd/dist.py) on CPU
tensors with a test double of the engine: slices, padding of ragged batches, the single all-gather,
and that every rank ends with the full, correctly ordered result."""

from __future__ import annotations

import os
import random
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, ret) -> None:
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_tensor_engine import FakeTensorEngine, _ints, _rows
        from oracle import oracle
        from protocols.distributed_keygen_amd import dist as mxdist, limbs as L

        eng = FakeTensorEngine()
        rng = random.Random(99)            # same inputs on every rank (replicated control)
        # --- shared-modulus modexp, ragged batch (7 rows over 2 ranks)
        mod = rng.getrandbits(200) | (1 << 199) | 1
        exp = rng.getrandbits(190)
        bases = [rng.randrange(mod) for _ in range(7)]
        got = mxdist.sharded_powmod_shared(eng, _rows(bases, L.limbs_for(mod)), mod, exp)
        assert _ints(got) == [pow(b, exp, mod) for b in bases]
        # --- partial decryptions modulo n^2 (N-adic kernel entry), ragged batch
        n_small = rng.getrandbits(100) | (1 << 99) | 1
        cs = [rng.randrange(n_small * n_small) for _ in range(5)]
        got = mxdist.sharded_powmod_nsquare(eng, _rows(cs, L.limbs_for(n_small * n_small)), n_small, exp)
        assert _ints(got) == [pow(c, exp, n_small * n_small) for c in cs]
        # --- per-candidate modexp, 3 candidates x 5 bases over 2 ranks
        mods = [rng.getrandbits(131) | (1 << 130) | 1 for _ in range(3)]
        exps = [rng.getrandbits(129) for _ in mods]
        flat = [rng.randrange(m) for m in mods for _ in range(5)]
        got = mxdist.sharded_powmod_multi(eng, _rows(flat, 5), mods, exps, 5)
        assert _ints(got) == [pow(b, exps[k // 5], mods[k // 5]) for k, b in enumerate(flat)]
        # --- sieve verdict bytes
        primes = oracle.small_prime_list(200)
        cands = [rng.getrandbits(68) | 1 for _ in range(9)]
        got = mxdist.sharded_sieve(eng, _rows(cands, 3), primes)
        assert [bool(x) for x in got.tolist()] == [oracle.small_prime_divisors_test(primes, c) for c in cands]
        # --- recombination over a ragged batch, with one inconsistent ciphertext
        from protocols.distributed_keygen_amd import synthetic

        key = synthetic.make_key(64, 3, 1, kappa=20)
        r2 = random.Random(5)
        msgs = [3, 1, 4, 1, 5]
        cts = [synthetic.encrypt(key, m, r2) for m in msgs]
        parts = [[oracle.partial_decrypt(c, key.n, i, key.degree, key.n_fac, key.shares[i]) for c in cts] for i in (1, 2, 3)]
        parts[1][3] = (parts[1][3] + 1) % key.n_square
        limbs2 = L.limbs_for(key.n_square)
        pt = torch.stack([_rows(p, limbs2) for p in parts])
        m_t, st = mxdist.sharded_combine(eng, pt, key.n, key.theta_inv)
        assert st.tolist() == [0, 0, 0, 1, 0]
        assert [m for m, s in zip(_ints(m_t), st.tolist()) if not s] == [3, 1, 4, 5]
        # --- the v-calculation of a keygen round: 5 candidates x 12 generators, keep 4, over 2 ranks (ragged)
        cm = [rng.getrandbits(130) | (1 << 129) | 1 for _ in range(5)]
        ce = [rng.getrandbits(128) for _ in cm]
        gens = [rng.randrange(m) for m in cm for _ in range(12)]
        gens[12:24] = [0] * 12                                   # a candidate with no Jacobi-1 generator at all
        v_t, cnt = mxdist.sharded_biprime_v(eng, _rows(gens, 5), cm, ce, 12, 4)
        want_v, want_c = [], []
        for c, (m, e) in enumerate(zip(cm, ce)):
            kept = [g for g in gens[c * 12 : (c + 1) * 12] if oracle.jacobi_symbol(g, m) == 1][:4]
            want_c.append(len(kept))
            want_v += [pow(g, e, m) for g in kept] + [pow(0, e, m)] * (4 - len(kept))
        assert cnt.tolist() == want_c and want_c[1] == 0 and _ints(v_t) == want_v
        # --- a caller holding only its shard: single-GPU operator + all_gather_rows
        lo, hi = mxdist.shard_bounds(7, dist.get_rank(), 2)
        local = eng.powmod_shared_t(_rows(bases[lo:hi], L.limbs_for(mod)), mod, exp)
        assert _ints(mxdist.all_gather_rows(local, 7)) == [pow(b, exp, mod) for b in bases]
        # --- fewer rows than ranks: rank 1's shard is empty (padding an empty shard must not raise
        # and leave the other ranks hanging in the collective)
        lo, hi = mxdist.shard_bounds(1, dist.get_rank(), 2)
        assert (lo, hi) == ((0, 1) if dist.get_rank() == 0 else (1, 1))
        local = eng.powmod_shared_t(_rows(bases[lo:hi], L.limbs_for(mod)), mod, exp) if hi > lo else _rows([], L.limbs_for(mod))
        assert _ints(mxdist.all_gather_rows(local, 1)) == [pow(bases[0], exp, mod)]
        # --- biprimality vote
        m0 = (1 << 100) + 277
        v = torch.stack([_rows([5, 6, 7, 9, 2, 4], 4), _rows([5, m0 - 6, 8, 9, 2, 4], 4), _rows([1] * 6, 4)]).reshape(3, 3, 2, 4)
        votes = mxdist.sharded_biprime_vote(eng, v, [m0, m0, m0])
        assert votes.tolist() == [[1, 1], [0, 1], [1, 1]]
        assert mxdist.shard_bounds(7, 0, 2) == (0, 4) and mxdist.shard_bounds(7, 1, 2) == (4, 7)
        # --- shard-only inputs: a rank packs and passes ONLY its slice, `total` names the whole
        me = dist.get_rank()
        lo, hi = mxdist.shard_bounds(7, me, 2)
        got = mxdist.sharded_powmod_shared(eng, _rows(bases[lo:hi], L.limbs_for(mod)), mod, exp, total=7)
        assert _ints(got) == [pow(b, exp, mod) for b in bases]
        lo, hi = mxdist.shard_bounds(5, me, 2)
        got = mxdist.sharded_powmod_nsquare(eng, _rows(cs[lo:hi], L.limbs_for(n_small * n_small)), n_small, exp, total=5)
        assert _ints(got) == [pow(c, exp, n_small * n_small) for c in cs]
        lo, hi = mxdist.shard_bounds(3, me, 2)
        got = mxdist.sharded_powmod_multi(eng, _rows(flat[lo * 5 : hi * 5], 5), mods[lo:hi], exps[lo:hi], 5, total=3)
        assert _ints(got) == [pow(b, exps[k // 5], mods[k // 5]) for k, b in enumerate(flat)]
        lo, hi = mxdist.shard_bounds(9, me, 2)
        got = mxdist.sharded_sieve(eng, _rows(cands[lo:hi], 3), primes, total=9)
        assert [bool(x) for x in got.tolist()] == [oracle.small_prime_divisors_test(primes, c) for c in cands]
        lo, hi = mxdist.shard_bounds(5, me, 2)
        m_t, st = mxdist.sharded_combine(eng, pt[:, lo:hi].contiguous(), key.n, key.theta_inv, total=5)
        assert st.tolist() == [0, 0, 0, 1, 0] and [m for m, s_ in zip(_ints(m_t), st.tolist()) if not s_] == [3, 1, 4, 5]
        v_t2, cnt2 = mxdist.sharded_biprime_v(eng, _rows(gens[lo * 12 : hi * 12], 5), cm[lo:hi], ce[lo:hi], 12, 4, total=5)
        assert cnt2.tolist() == want_c and _ints(v_t2) == want_v
        lo, hi = mxdist.shard_bounds(3, me, 2)
        votes = mxdist.sharded_biprime_vote(eng, v[:, lo:hi].contiguous(), [m0] * (hi - lo), total=3)
        assert votes.tolist() == [[1, 1], [0, 1], [1, 1]]
        # one unit over two ranks: rank 1 passes an EMPTY shard to every operator and still joins the collective
        lo, hi = mxdist.shard_bounds(1, me, 2)
        got = mxdist.sharded_powmod_shared(eng, _rows(bases[lo:hi], L.limbs_for(mod)), mod, exp, total=1)
        assert _ints(got) == [pow(bases[0], exp, mod)]
        got = mxdist.sharded_powmod_multi(eng, _rows(flat[lo * 5 : hi * 5], 5), mods[lo:hi], exps[lo:hi], 5, total=1)
        assert _ints(got) == [pow(b, exps[0], mods[0]) for b in flat[:5]]
        m_t, st = mxdist.sharded_combine(eng, pt[:, lo:hi].contiguous(), key.n, key.theta_inv, total=1)
        assert st.tolist() == [0] and _ints(m_t) == [3]
        v_t2, cnt2 = mxdist.sharded_biprime_v(eng, _rows(gens[lo * 12 : hi * 12], 5), cm[lo:hi], ce[lo:hi], 12, 4, total=1)
        assert cnt2.tolist() == want_c[:1] and _ints(v_t2) == want_v[:4]
        votes = mxdist.sharded_biprime_vote(eng, v[:, lo:hi].contiguous(), [m0] * (hi - lo), total=1)
        assert votes.tolist() == [[1, 1]]
        # a shard of the wrong size is refused before the collective (every rank raises: nobody is left waiting)
        try:
            mxdist.sharded_sieve(eng, _rows(cands, 3), primes, total=9)
            raise AssertionError("full input with total= accepted")
        except ValueError:
            pass
        ret[rank] = "ok"
    except Exception as exc:  # surfaced by the parent
        import traceback

        ret[rank] = "FAIL: " + "".join(traceback.format_exception(type(exc), exc, exc.__traceback__))[-1500:]
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_ops_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
        assert all(not p.is_alive() for p in procs), "gloo workers hung"
        assert dict(ret) == {0: "ok", 1: "ok"}, dict(ret)


def test_single_process_falls_through():
    sys.path.insert(0, str(ROOT / "tests"))
    from fake_tensor_engine import FakeTensorEngine, _ints, _rows
    from protocols.distributed_keygen_amd import dist as mxdist

    got = mxdist.sharded_powmod_shared(FakeTensorEngine(), _rows([2, 3, 4], 2), 1000003, 5)
    assert _ints(got) == [32, 243, 1024]
