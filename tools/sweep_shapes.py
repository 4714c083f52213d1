#!/usr/bin/env python3
"""Duration of ONE powmod_nsquare launch for every launch shape (limbs per lane x wavefronts per group) over a
range of batch sizes — the measurements the library's automatic choice (csrc/mx_capi_n2.hip: n2_estimate) is
calibrated against.  usage: sweep_shapes.py [key_length ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic  # noqa: E402

eng = Engine()
SHAPES = [(3, 2, 1), (9, 2, 1), (18, 2, 1), (9, 1, 1), (18, 1, 1), (9, 2, 2), (18, 2, 2)]    # (L, wavefronts, time-slice knob)
for key_length in [int(a) for a in sys.argv[1:]] or [2048]:
    key = synthetic.make_key(key_length, 3, 1)
    own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
    exp = key.exponent(own)
    n, n2 = key.n, key.n_square
    limbs2 = L.limbs_for(n2)
    sizes = [1, 64, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 10000, 12288, 16384, 20000, 32768]
    if key_length > 2048:
        sizes = [1, 64, 256, 512, 1024, 2048, 4096, 8192, 16384]
    cts = synthetic.random_ciphertexts(key, max(sizes), seed=7)
    c_all = eng.to_device(L.pack(cts, limbs2))
    print(f"key_length {key_length}: N {n.bit_length()} bits, exponent {exp.bit_length()} bits; ms per launch (one launch at a time, 1 segment)")
    print(f"{'batch':>7s} " + " ".join((f"L{l}x{w}w" + ("ts" if ts == 2 else "")).rjust(10) for l, w, ts in SHAPES) + "   auto-choice")
    ref = None
    for b in sizes:
        row = []
        for lpl, wpg, ts in SHAPES:
            eng.set_limbs_per_lane(lpl)
            eng.set_wavefronts_per_group(wpg)
            eng.debug_knob("n2_timeslice", ts)
            if ts == 2 and b < 512:
                row.append(None)
                continue
            try:
                eng.nsquare_launch_shape(n.bit_length(), b)
            except Exception:
                row.append(None)
                continue
            seg = 1 if ts == 1 else 0
            out = eng.powmod_nsquare_t(c_all[:b], n, exp, segments=seg)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                out = eng.powmod_nsquare_t(c_all[:b], n, exp, segments=seg)
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            if b in (64, 1024):
                got = L.unpack(eng.to_host(out))
                if ref is None or len(ref) != b:
                    ref = got
                    assert got[:4] == [pow(c, exp, n2) for c in cts[:4]]
                assert got == ref, (lpl, wpg, ts)
            row.append(best * 1e3)
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)
        eng.debug_knob("n2_timeslice", 0)
        k, l, w, blk, wv = eng.nsquare_launch_shape(n.bit_length(), b)
        fastest = min((t, s) for t, s in zip(row, SHAPES) if t is not None)
        t0 = time.perf_counter()
        eng.powmod_nsquare_t(c_all[:b], n, exp)
        torch.cuda.synchronize()
        t_auto = (time.perf_counter() - t0) * 1e3
        print(f"{b:7d} " + " ".join(("%10.2f" % t) if t is not None else "         -" for t in row)
              + f"   L{l}x{wv}w (K={k})  auto {t_auto:.2f} ms  fastest L{fastest[1][0]}x{fastest[1][1]}w{'ts' if fastest[1][2] == 2 else ''}  {b / fastest[0]:.0f} k/s")
