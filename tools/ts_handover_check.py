#!/usr/bin/env python3
"""Hand-overs of a time-sliced launch under the worst timing: as many resident pairs as groups (or more), so that a free
pair is already waiting when a group's next unit is published and reads what the previous pair stored the moment the
entry appears.  Repeats one configuration and counts rows that differ from the plain one-wavefront launch, plus the state
the queues are left in (every level granted and written exactly once per group).  The checker is tests/handover_check.py;
`pytest -m gpu` runs it on fixed shapes (tests/test_gpu_handover.py), this tool on any.

usage: ts_handover_check.py <batch> <limbs_per_lane 9|18> <resident per CU> <units per group> [repeats] [key_length]
A/B: tools/prove_handover_guard.sh (build with -DMX_DEV_TS_COMPILER_RELEASE, run through MX_LIBRARY) — the compiler's own
sequence for the release store (no wait for the L2 write-back before the flag) loses groups."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import handover_check as H

from protocols.distributed_keygen_amd import Engine

b, lpl, r, seg = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 12
key_length = int(sys.argv[6]) if len(sys.argv) > 6 else 2048
wrong, bad_queues, _ = H.check(Engine(), key_length, b, lpl, r, seg, reps, log=lambda s: print(s, flush=True))
sys.exit(1 if wrong or bad_queues else 0)
