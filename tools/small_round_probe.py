#!/usr/bin/env python3
"""Where the time of a SMALL keygen round's v-calculation goes (1 .. 25 survivors: the reference's batch sizes): each device
stage on its own, lone launches on an idle GPU — Jacobi filter (head / tail / all 160 in one launch), selection, modexps.
usage: small_round_probe.py [key_length]"""
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from protocols.distributed_keygen_amd import Engine, limbs as L

eng = Engine()
rng = random.Random(11)
key_length = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
bits = key_length + 3
limbs = L.limbs_for_bits(bits)


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3, out


for cands in (1, 2, 5, 20):
    mods = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(cands)]
    exps = [rng.getrandbits(bits - 2) | (1 << (bits - 3)) for _ in mods]
    g = [rng.randrange(m) for m in mods for _ in range(160)]
    g_t = eng.to_device(L.pack(g, limbs))
    mods_op = (eng.to_device(L.pack(mods, limbs)), bits)
    exps_op = (eng.to_device(L.pack(exps, L.limbs_for_bits(bits - 2))), bits - 2)
    j_t = torch.zeros(cands * 160, dtype=torch.int8, device=eng.device)
    t_head, _ = timed(lambda: eng.jacobi_t(g_t, mods_op, 160, out_t=j_t, first=0, count=104))
    t_tail, _ = timed(lambda: eng.jacobi_t(g_t, mods_op, 160, out_t=j_t, first=104, count=56))
    t_all, _ = timed(lambda: eng.jacobi_t(g_t, mods_op, 160, out_t=j_t))
    t_sel, (sel_t, cnt_t) = timed(lambda: eng.select_first_t(g_t, j_t, 160, 40))
    t_pow, _ = timed(lambda: eng.powmod_multi_t(sel_t, mods_op, exps_op, 40))
    t_v, _ = timed(lambda: eng.biprime_v_t(g_t, mods_op, exps_op, 160, 40))
    print(f"key_length {key_length}, {cands:2d} candidates: jacobi head(104) {t_head:.2f} ms, tail(56) {t_tail:.2f}, all 160 in one launch {t_all:.2f}; "
          f"select {t_sel:.2f}; 40 modexps each {t_pow:.2f} (form {eng.generic_launch_form(bits, 40 * cands, cands)}); biprime_v_t {t_v:.2f} ms", flush=True)
