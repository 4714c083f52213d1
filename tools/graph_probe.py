#!/usr/bin/env python3
"""Is a small biprimality-test step bound by kernel launches?  One step (Jacobi filter -> selection -> 40 modexps per
candidate -> verdict) for 256 candidates at key_length 1024: host time to ENQUEUE a step, steps/s with k lanes in flight
enqueued kernel by kernel, and the same steps captured once per lane as HIP graphs and replayed (developer probe)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocols.distributed_keygen_amd import configure_hw_queues

configure_hw_queues(16)
import torch

sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench

key_length = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cands = int(sys.argv[2]) if len(sys.argv) > 2 else 256
from protocols.distributed_keygen_amd import Engine

eng = Engine()
wl = bench.BiprimeWorkload(eng, key_length, 5, cands, seed=1)
for lanes in (1, 2, 4, 6):
    eng.set_limbs_per_lane(eng.geometry(wl.mod_bits, cands * wl.KEEP * lanes, cands * lanes)[1])
    eng.set_priority_aux(lanes > 1)
    wl.make_lanes(lanes, None, 1)
    for k in range(2 * lanes):
        wl.step(k, None)
    torch.cuda.synchronize()
    steps = 12 * lanes
    t0 = time.perf_counter()
    for k in range(steps):
        wl.step(k, None)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"lanes {lanes}: eager {cands * wl.KEEP * steps / dt / 1e6:.2f} M modexps/s, {dt / steps * 1e3:.2f} ms/step, host enqueue {t_enq / steps * 1e3:.2f} ms/step", flush=True)
    # the same step of every lane as a graph
    eng.set_priority_aux(False)
    graphs = []
    try:
        for k in range(lanes):
            ln = wl.lanes[k]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=ln["stream"] if lanes > 1 else None):
                v_t, _ = eng.biprime_v_t(wl.g_t, wl.mods_op, wl.exps_op, wl.GENS, wl.KEEP)
                ln["v_all"][wl.index - 1].view(-1, wl.limbs).copy_(v_t)
                eng.biprime_verdict_t(ln["v_all"], wl.mods_op, pass_t=ln["verdict"])
            graphs.append(g)
        torch.cuda.synchronize()
        for g in graphs:
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            graphs[k % lanes].replay()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"lanes {lanes}: graph {cands * wl.KEEP * steps / dt / 1e6:.2f} M modexps/s, {dt / steps * 1e3:.2f} ms/step, host enqueue {t_enq / steps * 1e3:.3f} ms/step; {wl.verify(3)}", flush=True)
    except Exception as exc:
        print(f"lanes {lanes}: graph capture failed: {type(exc).__name__}: {exc}", flush=True)
