#!/usr/bin/env python3
"""Time-sliced vs plain two-wavefront launches over batch sizes, resident workgroups per CU and segment counts
(developer probe behind the choice in csrc/mx_capi_n2.hip: n2_estimate).  usage: ts_probe.py [key_length]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic
eng = Engine()
key_length = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
key = synthetic.make_key(key_length, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp = key.exponent(own); n = key.n
limbs2 = L.limbs_for(key.n_square)
sizes = [int(v) for v in os.environ.get("TS_SIZES", "8192,9216,10000,10240,11264,12288,13312,14336,16384").split(",")]
if key_length > 2048 and "TS_SIZES" not in os.environ:
    sizes = [2048, 2304, 2560, 3072, 3584, 4096]
segs = [int(v) for v in os.environ.get("TS_SEGS", "2,3,4,6,8").split(",")]
cts = synthetic.random_ciphertexts(key, max(sizes), seed=7)
c_all = eng.to_device(L.pack(cts, limbs2))
def t(b, lpl, wpg, ts, seg):
    eng.set_limbs_per_lane(lpl); eng.set_wavefronts_per_group(wpg); eng.debug_knob("n2_timeslice", ts)
    out = eng.powmod_nsquare_t(c_all[:b], n, exp, segments=seg); torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); eng.powmod_nsquare_t(c_all[:b], n, exp, segments=seg); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out
for b in sizes:
    plain = {f"L{l}x{w}": t(b, l, w, 1, 1) for l, w in ((9, 2), (18, 2), (18, 1))}
    want = plain["L18x1"][1]
    assert all(torch.equal(v[1], want) for v in plain.values())
    plain = {k: v[0] for k, v in plain.items()}
    row = [f"b{b}: plain " + " ".join(f"{k} {v:.1f}" for k, v in plain.items()) + " | ts"]
    best = (min(plain.values()), "plain")
    for lpl, rs in ((9, (1, 2, 3)), (18, (1,))):
        for r in rs:
            for seg in segs:
                try:
                    v, out = t(b, lpl, 2, 16 + r, seg)
                except Exception:          # no time-sliced instance at this geometry in the library as built
                    continue
                bad = int((out != want).any(dim=1).sum())                # bit-identical to the plain launches?
                best = min(best, (v, f"L{lpl}r{r}s{seg}"))
                row.append(f"L{lpl}r{r}s{seg} {v:.1f}" + (f" ({bad} WRONG ROWS)" if bad else ""))
    eng.debug_knob("n2_timeslice", 0); eng.set_limbs_per_lane(0); eng.set_wavefronts_per_group(0)
    auto, out = t(b, 0, 0, 0, 0)
    assert torch.equal(out, want)
    print(" ".join(row), f"|| best {best[1]} {best[0]:.1f} vs plain {min(plain.values()):.1f}; library's choice {auto:.1f}", flush=True)
