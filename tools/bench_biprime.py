#!/usr/bin/env python3
"""Supplementary measurement (superseded by `bench.py --workload biprime`, kept for the stage-by-stage
figures): biprimality-test modexps/s on one GPU.

configs[1]/[3] shape of BASELINE.json: `cands` candidate moduli (key_length bits + 2..5), 40
Jacobi-1 bases each, party-1 exponent (N - p_1 - q_1 + 1)/4 (distributed_keygen.py:1094).
Prints one JSON line with the GPU rate, the Jacobi-filter rate, the sieve rate and a gmpy2/libgmp
single-core figure for the same modexp.
"""
import argparse, json, os, random, subprocess, sys, tempfile, time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key-length", type=int, default=2048)
    ap.add_argument("--parties", type=int, default=3)
    ap.add_argument("--cands", type=int, default=1024)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--limbs-per-lane", type=int, default=0, help="lane geometry 9|18, 0 = the library's choice")
    args = ap.parse_args()
    import torch
    import sympy
    from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

    eng = Engine()
    eng.set_limbs_per_lane(args.limbs_per_lane)
    rng = random.Random(args.key_length)
    half = args.key_length // 2
    primes = [int(p) for p in sympy.primerange(3, 2001)]
    shares, mods = [], []
    while len(mods) < args.cands:                      # candidates that survive the sieve, as in DK:1288-1292
        cand = [synthetic.candidate_shares(rng, args.parties, half) for _ in range(args.cands * 8)]
        cm = [sum(p) * sum(q) for p, q in cand]
        keep = eng.sieve_batch(cm, primes)
        for sh, m, bad in zip(cand, cm, keep):
            if not bad and len(mods) < args.cands:
                shares.append(sh)
                mods.append(m)
    exps = [(m - p[0] - q[0] + 1) // 4 for m, (p, q) in zip(mods, shares)]
    limbs = L.limbs_for_bits(max(m.bit_length() for m in mods))
    g_all = [[rng.randrange(m) for _ in range(160)] for m in mods]
    # Jacobi filter on the device
    g_t = eng.to_device(L.pack([g for row in g_all for g in row], limbs))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        j_t = eng.jacobi_t(g_t, mods, 160)
    torch.cuda.synchronize(); jac_s = (time.perf_counter() - t0) / args.steps
    jac = j_t.cpu().numpy().reshape(args.cands, 160)
    kept = []
    for c in range(args.cands):
        row = [g_all[c][k] for k in range(160) if jac[c, k] == 1][:40]
        assert len(row) >= 20
        kept.append((row * 2)[:40])
    b_t = eng.to_device(L.pack([g for row in kept for g in row], limbs))
    out_t = torch.empty_like(b_t)
    eng.powmod_multi_t(b_t, mods, exps, 40, out_t=out_t)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.powmod_multi_t(b_t, mods, exps, 40, out_t=out_t)
    torch.cuda.synchronize(); pm_s = (time.perf_counter() - t0) / args.steps
    rows = L.unpack(eng.to_host(out_t[:80]))
    assert rows == [pow(kept[c][k], exps[c], mods[c]) for c in range(2) for k in range(40)]
    assert int(jac[0, 5]) == sympy.jacobi_symbol(g_all[0][5], mods[0])
    # sieve
    c_t = eng.to_device(L.pack(mods * 16, limbs))
    eng.sieve_t(c_t, primes)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.sieve_t(c_t, primes)
    torch.cuda.synchronize(); sv_s = (time.perf_counter() - t0) / args.steps
    # CPU: same modexp on one core with the reference's engine
    job = {"mod": hex(mods[0]), "exp": hex(exps[0]), "bases": [hex(g) for g in kept[0]], "nprocs": 1, "seconds": 2.0}
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(job, f)
    cpu = None
    for py in ("/opt/conda/bin/python3.9", sys.executable):
        if os.path.exists(py):
            r = subprocess.run([py, str(ROOT / "oracle" / "cpu_baseline.py"), f.name], capture_output=True, text=True)
            if r.returncode == 0:
                cpu = json.loads(r.stdout.strip().splitlines()[-1])
                break
    n = args.cands * 40
    print(json.dumps({
        "workload": f"biprime test: {args.cands} candidates x 40 bases, key_length {args.key_length}, {args.parties} parties",
        "mod_bits": max(m.bit_length() for m in mods), "exp_bits": max(e.bit_length() for e in exps),
        "modexps_per_s": n / pm_s, "powmod_ms": pm_s * 1e3,
        "jacobi_per_s": args.cands * 160 / jac_s, "jacobi_ms": jac_s * 1e3,
        "sieve_candidates_per_s": len(mods) * 16 / sv_s, "sieve_ms": sv_s * 1e3, "sieve_primes": len(primes),
        "geometry": eng.geometry(max(m.bit_length() for m in mods), args.cands * 40, args.cands),   # of THIS launch (batch, groups)
        "cpu_single_core_modexps_per_s": None if cpu is None else cpu["rate_single_core"],
        "cpu_engine": None if cpu is None else cpu["engine_desc"],
    }))


if __name__ == "__main__":
    main()
