#!/usr/bin/env python3
"""Census of wrong rows of powmod_nsquare launches that overlap on several streams (developer probe, round 4).

Round 3 had a build that was bit-exact on one stream and returned wrong rows with four launches in flight
(DESIGN.md §9 1b).  This script reproduces the situation for any build of the library (MX_LIBRARY=<variant>.so,
tools/build_variant.py) and says WHAT is wrong: how many rows per stream, whether whole wavefronts or single rows,
which workgroups (XCD = workgroup index mod 8), and — for builds with -DMX_DEV_PRIVATE_PAD_WORDS=n — whether a wavefront's
private scratch was overwritten while it ran, and by whom (the pad pattern names its writer).

  python tools/concurrency_census.py [--rows 10000] [--streams 4] [--queues 16] [--segments 0] [--shape 18,1]
                                     [--one-stream] [--reps 2] [--label text]

--queues N sets GPU_MAX_HW_QUEUES before the runtime initialises; --one-stream enqueues the same launches on ONE
stream (no overlap: the control).  Expected rows come from CPython pow on the host cores, cached in --cache.
"""
import argparse
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10000)
ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--queues", type=int, default=0)
ap.add_argument("--segments", type=int, default=-1, help="-1: the engine's default (library's choice)")
ap.add_argument("--shape", default="18,1")
ap.add_argument("--one-stream", action="store_true")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--label", default="")
ap.add_argument("--cache", default="/tmp/census_want")
ap.add_argument("--timeslice", type=int, default=0)
ap.add_argument("--knob", action="append", default=[], metavar="NAME=VALUE", help="Engine.debug_knob overrides, e.g. n2_friendly_1w=1")
args = ap.parse_args()
if args.queues:
    os.environ["GPU_MAX_HW_QUEUES"] = str(args.queues)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes  # noqa: E402
import multiprocessing as mp  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from protocols.distributed_keygen_amd import Engine, _lib, limbs as L, synthetic  # noqa: E402

if os.environ.get("MX_LIBRARY"):                     # variant builds of other sources may lack the newest entry points
    probe = ctypes.CDLL(os.environ["MX_LIBRARY"])
    for name in [n for n in _lib.SYMBOLS if not hasattr(probe, n)]:
        del _lib.SYMBOLS[name]

key = synthetic.make_key(2048, 3, 1)
own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
exp, n, n2 = key.exponent(own), key.n, key.n_square
limbs2 = L.limbs_for(n2)
cts = synthetic.random_ciphertexts(key, args.rows, seed=11)
cache = f"{args.cache}_{args.rows}.npy"
if os.path.exists(cache):
    want_rows = np.load(cache)
else:
    t0 = time.perf_counter()
    with mp.Pool(16) as pool:
        want = pool.starmap(pow, [(c, exp, n2) for c in cts], chunksize=16)
    want_rows = L.pack(want, limbs2)
    np.save(cache, want_rows)
    print(f"# expected rows: {args.rows} x pow() on the host cores in {time.perf_counter() - t0:.1f} s", flush=True)

eng = Engine()
lib = eng.lib
has_pad = hasattr(lib, "mx_debug_pad_faults")
lpl, wpg = (int(x) for x in args.shape.split(","))
eng.set_limbs_per_lane(lpl)
eng.set_wavefronts_per_group(wpg)
if args.timeslice:
    eng.debug_knob("n2_timeslice", args.timeslice)
for kv in args.knob:
    eng.debug_knob(kv.partition("=")[0], int(kv.partition("=")[2]))
seg = None if args.segments < 0 else args.segments
c_t = eng.to_device(L.pack(cts, limbs2))
want_t = eng.to_device(want_rows)
K = eng.nsquare_launch_shape(n.bit_length(), args.rows)[0]
per_wave = 64 // K
tag = f"{args.label or os.path.basename(os.environ.get('MX_LIBRARY', 'shipped'))} shape {lpl}x{wpg}w rows {args.rows} segments {args.segments} queues {args.queues or 'default'}"


def pad_faults(where):
    if not has_pad:
        return
    buf = (ctypes.c_uint32 * (7 * 32))()
    lib.mx_debug_pad_faults.restype = ctypes.c_int
    cnt = lib.mx_debug_pad_faults(buf, 32)
    print(f"   pad faults {where}: {cnt}", flush=True)
    for k in range(min(cnt, 8)):
        t, blk, lane, idx, w, g, fl = (buf[7 * k + j] for j in range(7))
        x = g ^ ((idx >> 6) * 0x9E3779B1 & 0xFFFFFFFF)
        print(f"      tag {t:#04x} workgroup {blk} lane {lane} word {idx} (first*2+last {fl}): expected {w:#010x} found {g:#010x}"
              f"  -> decodes as tag {(x >> 24) & 0xFF:#04x} workgroup {(x >> 12) & 0xFFF} lane {(x >> 6) & 63} word&63 {x & 63}", flush=True)


def census(out_t, name):
    bad = (out_t != want_t).any(dim=1).nonzero().flatten().cpu().numpy()
    if len(bad) == 0:
        print(f"   {name}: 0 wrong rows of {args.rows}", flush=True)
        return 0
    waves = np.unique(bad // per_wave)
    full = sum(1 for w in waves if np.sum(bad // per_wave == w) == min(per_wave, args.rows - w * per_wave))
    xcd = np.bincount(waves % 8, minlength=8)
    print(f"   {name}: {len(bad)} WRONG rows of {args.rows} in {len(waves)} wavefronts ({full} of them entirely wrong); "
          f"wavefront index mod 8: {xcd.tolist()}; first wavefronts {waves[:10].tolist()}; first rows {bad[:8].tolist()}", flush=True)
    i = int(bad[0])
    got = L.unpack(eng.to_host(out_t[i : i + 1]))[0]
    w = L.unpack(want_rows[i : i + 1])[0]
    d = (got - w) % n2
    print(f"      row {i}: got < N^2: {got < n2}; (got - want) mod N = {'0' if d % n == 0 else 'nonzero'}; got == 0: {got == 0}", flush=True)
    return len(bad)


print(f"== {tag}", flush=True)
pad_faults("before (reset)")
out = eng.powmod_nsquare_t(c_t, n, exp, segments=seg)
torch.cuda.synchronize()
census(out, "single launch")
pad_faults("after the single launch")
streams = [torch.cuda.Stream() for _ in range(args.streams)]
total = 0
for rep in range(args.reps):
    outs, evs = [], []
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    t0 = time.perf_counter()
    for st in streams:
        use = streams[0] if args.one_stream else st
        use.wait_event(base)
        with torch.cuda.stream(use):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            outs.append(eng.powmod_nsquare_t(c_t, n, exp, segments=seg))
            e1.record()
            evs.append((e0, e1))
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    spans = ", ".join(f"{base.elapsed_time(a):.1f}-{base.elapsed_time(b):.1f}" for a, b in evs)
    print(f"  rep {rep}: {args.streams} launches {'on ONE stream' if args.one_stream else 'on ' + str(args.streams) + ' streams'} in {dt * 1e3:.1f} ms"
          f" (enqueued in {t_enq * 1e3:.1f} ms; start-end of each launch, ms: {spans})", flush=True)
    for k, o in enumerate(outs):
        total += census(o, f"launch {k}")
    pad_faults(f"after rep {rep}")
print(f"== {tag}: {total} wrong rows in total", flush=True)
