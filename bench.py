#!/usr/bin/env python3
"""Headline benchmark: modexps/sec (2048-bit N, mod N^2) — BASELINE.json's metric.

Workload (BASELINE.json configs[2], SURVEY.md §8d "C3"): one 3-party threshold-Paillier key,
key_length 2048, t = 1; a step = one party's pass over a batch of 10 000 ciphertexts:
  partial decryption   c^exp_i mod N^2     (paillier_shared_key.py:92 looped at distributed_keygen.py:463-466)
  share recombination  of the 3 partials   (paillier_shared_key.py:95-127 looped at distributed_keygen.py:510-515)
Inputs (ciphertext rows, the other parties' partial rows) are resident in HBM before the timed
region.  With N GPUs every rank processes its own 10 000-ciphertext batch (weak scaling, batches
are independent) and the partial-decryption rows are all-gathered over RCCL, which is the only
exchange step the path has.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)

Prints ONE JSON line (rank 0).  `roofline` and `cpu_baseline` are described in DESIGN.md §6.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# The HIP runtime multiplexes streams onto 4 hardware queues by default; with more than 3 steps in
# flight two streams then share a queue and serialise (measured: 4 streams 203 k modexps/s with 4
# queues, 246-272 k with 8; tools/ab_queues.sh).  Must be set before the runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s
# measured on MI355X: v_mad_u64_u32, 8 waves/SIMD, 2.085 ns per wave-instruction per SIMD
# (profiles/r01_ubench_valu_rates.txt)  ->  1024 SIMDs * 64 lanes / 2.085 ns
VALU_MAC_PEAK = 1024 * 64 / 2.085e-9
# HBM bytes per powmod launch of the default workload, from the rocprofv3 --pmc passes committed in
# profiles/r01_bench_single_stream_{wide,narrow}_rocprof_summary.txt: (2 x FETCH_SIZE + WRITE_SIZE) KiB
# (FETCH_SIZE counts half of wide coalesced reads on gfx950, MI355X guide).  It is the table of odd
# powers (64 pairs per ciphertext, 72 slots in all): ~0.4-0.5 GB written and ~2.7 GB of coalesced
# look-ups per 10 000 modexps.  Keyed by limbs per lane (narrow, wide geometry).
MEASURED_TRAFFIC_DEFAULT = {9: (2 * 1320544 + 410039) * 1024, 18: (2 * 1414320 + 501583) * 1024}
# VALU wave-instructions one powmod_n2_kernel launch of the default workload issues (SQ_INSTS_VALU of
# the same profile), and the issue peak: 1024 SIMDs, one VALU instruction per 4 cycles at 2.4 GHz.
MEASURED_VALU_INSTS_DEFAULT = {9: 2.2432e10, 18: 1.8393e10}     # by limbs per lane (narrow, wide geometry)
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4


def parse() -> argparse.Namespace:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 48 timed steps: with 4 steps in flight the last round drains a partly empty machine, which costs
    # ~10 % of a 12-step run (267 k) and ~2 % of a 48-step one (292-300 k, the sustained rate)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=10000, help="ciphertexts per step per GPU")
    ap.add_argument("--key-length", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    ap.add_argument("--check", type=int, default=6, help="elements verified against CPython pow after timing")
    ap.add_argument("--streams", type=int, default=0,
                    help="independent steps (10k-ciphertext batches) kept in flight, one HIP stream each; "
                         "1 = strictly one batch at a time; 0 = automatic: the first of 4, 5, 6, 7, 8, 3 that "
                         "divides --steps (every stream then runs the same number of steps), else 4")
    ap.add_argument("--limbs-per-lane", type=int, default=-1,
                    help="lane geometry 9|18, 0 = library heuristic; default: 18 (wide) when at least 4 steps "
                         "are in flight and the modulus has >= 2048 bits")
    ap.add_argument("--generic-modulus", action="store_true",
                    help="time mx_powmod_shared on the modulus N^2 instead of mx_powmod_nsquare (pairs modulo N)")
    return ap.parse_args()


def cpu_baseline(key, exp: int, ciphertexts, seconds: float) -> dict:
    """Times the reference's CPU engine (gmpy2 -> libgmp) on all host cores, on a bounded sample."""
    sample = ciphertexts[:64]
    job = {
        "mod": hex(key.n_square), "exp": hex(exp), "bases": [hex(c) for c in sample],
        "nprocs": os.cpu_count() or 1, "seconds": seconds,
    }
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(job, f)
        path = f.name
    script = str(ROOT / "oracle" / "cpu_baseline.py")
    tried = []
    for py in ("/opt/conda/bin/python3.9", sys.executable):
        if not os.path.exists(py):
            continue
        try:
            r = subprocess.run([py, script, path], capture_output=True, text=True, timeout=seconds * 6 + 120)
            if r.returncode == 0 and r.stdout.strip():
                res = json.loads(r.stdout.strip().splitlines()[-1])
                if py != sys.executable and res["engine"] != "gmpy2":
                    tried.append(f"{py}: no gmpy2")
                    continue
                os.unlink(path)
                return {
                    "value": res["rate_all_cores"], "unit": "modexps/s", "cores": res["cores"],
                    "kind": "reference",
                    "engine": res["engine_desc"],
                    "single_core_value": res["rate_single_core"],
                    "sample": (f"{res['modexps_timed']} modexps in {res['wall_s']:.1f} s wall on {res['cores']} processes, "
                               f"same modulus/exponent, first {len(sample)} ciphertexts of the batch cycled; engine = the "
                               "routine the reference's pow_mod dispatches to (gmpy2.powmod -> libgmp mpz_powm)"),
                }
            tried.append(f"{py}: rc={r.returncode} {r.stderr[-200:]}")
        except Exception as exc:  # pragma: no cover - measurement plumbing
            tried.append(f"{py}: {exc}")
    os.unlink(path)
    return {"value": None, "unit": "modexps/s", "cores": 0, "kind": "reference", "sample": "failed: " + "; ".join(tried)}


def main() -> None:
    args = parse()
    # stdout carries exactly ONE line, the JSON result.  Libraries write banners to the C-level stdout
    # (RCCL prints its version block when the first communicator is created, and stdio would flush it
    # after Python's own output), so file descriptor 1 is pointed at stderr for the whole run and the
    # result goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one process per GPU)")
    # one process per GPU; MX_BENCH_BACKEND=gloo lets several ranks share one GPU (single-GPU smoke
    # test of the multi-rank code path; RCCL refuses two ranks on one device)
    backend = os.environ.get("MX_BENCH_BACKEND", "nccl")
    local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    # MX_BENCH_FORCE_DIST=1 runs the process-group code path (RCCL init, all-gather, barrier) with a
    # single rank: the only way to exercise it on a one-GPU box
    force_dist = world == 1 and os.environ.get("MX_BENCH_FORCE_DIST") == "1"
    if force_dist:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if world > 1 or force_dist:
        import torch.distributed as dist  # type: ignore

        if backend == "nccl":
            # the all-gather kernels must find wavefront slots on a GPU that the modexp launches keep
            # full: give RCCL's stream the high-priority queue (falls back if the option is unavailable)
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                dist.init_process_group("nccl", pg_options=opts, device_id=torch.device("cuda", local_rank))
            except (AttributeError, TypeError):
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from protocols.distributed_keygen_amd import build as _build

    if _build.needs_build():         # missing or stale library: compile it (local rank 0), never a CPU path
        if local_rank == 0:
            _build.build(force=False)
        if dist is not None:
            dist.barrier()
    from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

    eng = Engine(local_rank)
    if args.streams <= 0:
        args.streams = next((d for d in (4, 5, 6, 7, 8, 3) if args.steps % d == 0), min(4, max(1, args.steps)))
    # Lane geometry (mx_set_limbs_per_lane, part of the C ABI): the wide geometry issues 18 % fewer
    # instructions per modexp; it is selected when the launches in flight together over-subscribe its
    # 2048 wavefront slots, otherwise the library's per-launch heuristic decides.
    if args.limbs_per_lane >= 0:
        lpl = args.limbs_per_lane
    elif args.generic_modulus:
        lpl = 18 if args.streams >= 3 else 0
    else:
        wide_lanes = 1
        while wide_lanes * 29 * 18 < args.key_length + 8:
            wide_lanes *= 2
        wide_waves_in_flight = args.streams * args.batch * wide_lanes // 64
        lpl = 18 if (args.streams >= 4 and args.key_length >= 2048 and wide_waves_in_flight >= 2048) else 0
    eng.set_limbs_per_lane(lpl)
    key = synthetic.make_key(args.key_length, 3, 1)
    n, n2 = key.n, key.n_square
    parties = list(range(1, key.degree + 2))
    exps = {i: key.exponent(i) for i in parties}
    own = next((i for i in parties if exps[i] >= 0), parties[0])
    batch = args.batch
    limbs2 = L.limbs_for(n2)
    cts = synthetic.random_ciphertexts(key, batch, seed=synthetic.SEED + 17 * rank)
    c_t = eng.to_device(L.pack(cts, limbs2))

    # ---- setup (untimed): the other parties' partial decryptions, as they would arrive over the wire
    nstreams = max(1, args.streams)
    partials_t = torch.empty((len(parties), batch, limbs2), dtype=torch.int32, device=eng.device)
    for k, i in enumerate(parties):
        eng.powmod_shared_t(c_t, n2, abs(exps[i]), out_t=partials_t[k])
        if exps[i] < 0:  # paillier_shared_key.py:89-91 — c^-|e| = (c^|e|)^-1, inverted on the host here
            vals = L.unpack(eng.to_host(partials_t[k]))
            partials_t[k].copy_(eng.to_device(L.pack([pow(v, -1, n2) for v in vals], limbs2)))
    own_exp = abs(exps[own])
    own_in_t = c_t
    if exps[own] < 0:
        own_in_t = eng.to_device(L.pack([pow(c, -1, n2) for c in cts], limbs2))
    own_slot = parties.index(own)
    theta_inv = key.theta_inv
    # one set of buffers (and one engine workspace) per in-flight step
    lanes = []
    for k in range(nstreams):
        lanes.append({
            "eng": eng,          # one engine: its workspace is per stream, the per-key plans are shared
            "stream": torch.cuda.current_stream() if nstreams == 1 else torch.cuda.Stream(),
            "partials": partials_t if k == 0 else partials_t.clone(),
            "msg": torch.empty((batch, L.limbs_for(n)), dtype=torch.int32, device=eng.device),
            "status": torch.empty(batch, dtype=torch.uint8, device=eng.device),
            "gathered": torch.empty((world, batch, limbs2), dtype=torch.int32, device=eng.device) if dist is not None else None,
        })
    torch.cuda.synchronize()
    msg_t, status_t, gathered = lanes[0]["msg"], lanes[0]["status"], lanes[0]["gathered"]

    def step(timed: bool, k: int = 0) -> None:
        ln = lanes[k % nstreams]
        with torch.cuda.stream(ln["stream"]):
            if args.generic_modulus:
                ln["eng"].powmod_shared_t(own_in_t, n2, own_exp, out_t=ln["partials"][own_slot])
            else:
                ln["eng"].powmod_nsquare_t(own_in_t, n, own_exp, out_t=ln["partials"][own_slot])
            if dist is not None:
                dist.all_gather_into_tensor(ln["gathered"].view(-1), ln["partials"][own_slot].reshape(-1))
            ln["eng"].combine_t(ln["partials"], n, theta_inv, out_t=ln["msg"], status_t=ln["status"])

    def barrier() -> None:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # setup (untimed): one pass per lane allocates that lane's workspace and output buffers, so that no
    # allocation (a device synchronisation) can fall into the timed region even when --warmup is
    # smaller than the number of steps in flight; then the W warm-up steps proper
    for k in range(nstreams):
        step(False, k)
    barrier()
    for k in range(args.warmup):
        step(False, k)
    barrier()
    # HIP events around the modexp kernel itself, recorded by the library on the stream it launches
    # on (mx_profile): the same interval rocprofv3 --kernel-trace reports for that kernel
    eng.profile(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(True, k)
    barrier()
    elapsed = time.perf_counter() - t0
    eng.profile(False)
    kernel_total_ms, kernel_launches = eng.profile_collect()
    assert kernel_launches == args.steps, (kernel_launches, args.steps)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    powmod_ms = kernel_total_ms / kernel_launches

    # ---- verification (outside the timed region)
    for ln in lanes:
        assert int(ln["status"].sum().item()) == 0, "share recombination flagged ciphertexts as inconsistent"
    if dist is not None:
        assert torch.equal(gathered[rank], partials_t[own_slot]), "all-gather shard mismatch"
    check_note = "skipped"
    if rank == 0 and args.check > 0:
        # spot check with CPython big-int arithmetic (the definition of the reference's pow_mod /
        # PaillierSharedKey.decrypt, paillier_shared_key.py:92 and :115-125)
        idx = [0, batch - 1] + [(k * 7919) % batch for k in range(1, max(1, args.check - 1))]
        rows = eng.to_host(partials_t[own_slot][idx])
        msgs = L.unpack(eng.to_host(msg_t[idx]))
        allp = [L.unpack(eng.to_host(partials_t[k][idx])) for k in range(len(parties))]
        for j, e in enumerate(idx):
            base = cts[e] if exps[own] >= 0 else pow(cts[e], -1, n2)
            assert L.unpack(rows[j : j + 1])[0] == pow(base, own_exp, n2), f"partial decryption {e} differs from pow()"
            x = 1
            for k in range(len(parties)):
                x = x * allp[k][j] % n2
            assert (x - 1) % n == 0 and msgs[j] == (x - 1) // n * theta_inv % n, f"plaintext {e} differs"
        check_note = f"{len(idx)} elements bit-exact vs CPython pow; all {batch} combines divisible by N"

    if rank == 0:
        total_modexps = world * batch * args.steps
        s_limbs, e_bits = limbs2, own_exp.bit_length()
        alg_bytes = batch * (2 * 4 * s_limbs) + 4 * s_limbs + (e_bits + 7) // 8
        alg_macs = batch * (e_bits + -(-e_bits // 5) + 16) * (2 * s_limbs * s_limbs + s_limbs)
        achieved_gbs = alg_bytes / (powmod_ms * 1e-3) / 1e9
        # aggregate VALU rate of this GPU over the timed region (launches of different steps overlap
        # when several steps are in flight, so per-launch durations would under-state it)
        agg_mac_rate = alg_macs * args.steps / elapsed
        out = {
            "metric": "modexps/sec (2048-bit N, mod N^2)",
            "value": total_modexps / elapsed,
            "unit": "modexps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"C3: 3-party key_length={args.key_length} t=1, {batch} ciphertexts/GPU/step: "
                            "partial-decrypt c^exp mod N^2 + share-combine (BASELINE.json configs[2])",
                "batch_per_gpu": batch, "mod_bits": n2.bit_length(), "exp_bits": e_bits,
                "limbs_u32": s_limbs, "party": own, "parallelism": f"dp{world}", "steps_in_flight": nstreams,
                "geometry_K_L_W_blocks": list(eng.geometry(n2.bit_length()) if args.generic_modulus else eng.nsquare_geometry(n.bit_length(), batch)),
                "algorithm": "Montgomery modulo N^2" if args.generic_modulus else "N-adic pairs, two Montgomery passes modulo N per product",
                "verified": check_note,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": ("mx::powmod_kernel<%d,%d,29,true>" % tuple(eng.geometry(n2.bit_length())[:2])) if args.generic_modulus
                          else ("mx::powmod_n2_kernel<%d,%d,29>" % tuple(eng.nsquare_geometry(n.bit_length(), batch)[:2])),
                "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": MEASURED_TRAFFIC_DEFAULT[eng.nsquare_geometry(n.bit_length(), batch)[1]]
                if (batch == 10000 and args.key_length == 2048 and not args.generic_modulus) else None,
                "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profiles/ (not collected live)",
                "kernel_ms": powmod_ms, "concurrent_launches": nstreams,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "integer-VALU-bound path (north_star: no MFMA); the HBM fraction is reported as asked "
                        "(bytes of ONE launch over its own duration; `concurrent_launches` launches overlap), "
                        "the binding roof is the v_mad_u64_u32 issue rate below",
                "valu": {
                    "achieved": agg_mac_rate / 1e12, "peak": VALU_MAC_PEAK / 1e12,
                    "unit": "T 32x32-bit MAC/s", "frac": agg_mac_rate / VALU_MAC_PEAK,
                    "algorithmic_macs_per_launch": alg_macs,
                    "issue_utilization": (MEASURED_VALU_INSTS_DEFAULT[eng.nsquare_geometry(n.bit_length(), batch)[1]] * args.steps / elapsed / VALU_ISSUE_PEAK)
                    if (batch == 10000 and args.key_length == 2048 and not args.generic_modulus) else None,
                    "issue_utilization_basis": "SQ_INSTS_VALU per launch (profiles/: 2.24e10 narrow, 1.84e10 wide geometry) x "
                                               "launches / wall time, over 1024 SIMDs x 2.4 GHz / 4 cycles per VALU instruction",
                    "basis": "all launches of the timed region / wall time of the region, this GPU; the MAC count is "
                             "SURVEY.md 8(d)'s figure for schoolbook Montgomery modulo N^2 — a fraction above 1 means "
                             "the kernel needs fewer multiply-accumulates than that figure assumes (symmetric squaring, "
                             "half-size passes modulo N)",
                },
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(key, own_exp, [c if exps[own] >= 0 else pow(c, -1, n2) for c in cts[:64]], args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
