#!/usr/bin/env python3
"""Hand-overs of a time-sliced launch under the worst timing: as many resident pairs as groups (or more), so that a free
pair is already waiting when a group's next unit is published and reads what the previous pair stored the moment the
entry appears.  Repeats one configuration and counts rows that differ from the plain one-wavefront launch, plus the state
the queues are left in (every level granted and written exactly once per group).

usage: ts_handover_check.py <batch> <limbs_per_lane 9|18> <resident per CU> <units per group> [repeats]
A/B: tools/build_variant.py compiler_release -DMX_TS_COMPILER_RELEASE, then MX_LIBRARY=.../variants/compiler_release.so —
the compiler's own sequence for the release store (no wait for the L2 write-back before the flag) loses groups."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L, synthetic

b, lpl, r, seg = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 12
eng = Engine()
key = synthetic.make_key(2048, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp, n = key.exponent(own), key.n
limbs2 = L.limbs_for(key.n_square)
c = eng.to_device(L.pack(synthetic.random_ciphertexts(key, b, seed=7), limbs2))
eng.set_limbs_per_lane(18); eng.set_wavefronts_per_group(1); eng.debug_knob("n2_timeslice", 1)
want = eng.powmod_nsquare_t(c, n, exp, segments=1).clone(); torch.cuda.synchronize()
eng.set_limbs_per_lane(lpl); eng.set_wavefronts_per_group(2); eng.debug_knob("n2_timeslice", 16 + r)
k, l, _w, _blocks, _wf = eng.nsquare_launch_shape(n.bit_length(), b)
gpw = 64 // k
groups, nblocks = (b + gpw - 1) // gpw, (b + 2 * gpw - 1) // (2 * gpw)
table_bytes = (72 * 2 * l * nblocks * 128 * 4 + 255) // 256 * 256          # 8 + 2^(7-1) slots of the pair table (window 7)
wrong_total = 0
for it in range(reps):
    out = eng.powmod_nsquare_t(c, n, exp, segments=seg); torch.cuda.synchronize()
    bad = (out != want).any(dim=1).nonzero().flatten().cpu().numpy()
    sched = eng._ws[eng._stream_ptr()][table_bytes: table_bytes + (32 + groups * (seg - 1)) * 4].view(torch.int32).cpu().numpy()
    ok = list(sched[:seg]) == [sched[0]] + [groups] * (seg - 1) and sched[0] >= groups and list(sched[17:16 + seg]) == [groups] * (seg - 1)
    for lv in range(1, seg):
        ring = sched[32 + (lv - 1) * groups: 32 + lv * groups]
        ok = ok and sorted(ring.tolist()) == list(range(1, groups + 1))
    wrong_total += len(bad)
    print(f"run {it}: {len(bad)} wrong rows (groups {sorted(set(int(x) // gpw for x in bad))[:12]}); queues consistent: {ok}", flush=True)
print(f"batch {b}, L{l} K={k}, {r} workgroup(s) per CU, {seg} units per group, {reps} runs: {wrong_total} wrong rows")
sys.exit(1 if wrong_total else 0)
