#!/usr/bin/env python3
"""CPU baseline timer for bench.py (TEST/MEASUREMENT INFRASTRUCTURE — never imported by the product).

Times the reference's CPU engine for this path on the host cores: the reference calls
``pow_mod`` of tno.mpc.encryption_schemes.utils (distributed_keygen.py:1094,1097;
paillier_shared_key.py:92), which is ``gmpy2.powmod`` when gmpy2 is installed (README.md:43-47)
and CPython ``pow`` otherwise.  Engines, in order of preference:
  gmpy2      gmpy2.powmod (needs an interpreter that has gmpy2; this image: /opt/conda/bin/python3.9)
  libgmp     __gmpz_powm of the system libgmp.so.10 through ctypes — the routine gmpy2.powmod calls
  cpython    built-in pow
Runs standalone under any Python >= 3.8:   cpu_baseline.py job.json   -> one JSON line on stdout.
job.json: {"mod": hex, "exp": hex, "bases": [hex, ...], "nprocs": int, "seconds": float}
"""

from __future__ import annotations

import ctypes
import json
import multiprocessing as mp
import os
import sys
import time


def _engine():
    try:
        import gmpy2  # type: ignore

        def prep(x):
            return gmpy2.mpz(x)

        def powm(b, e, m):
            return gmpy2.powmod(b, e, m)

        return "gmpy2", f"gmpy2 {gmpy2.version()} / GMP {'.'.join(map(str, gmpy2.mp_version().split()[-1].split('.')))}", prep, powm, int
    except Exception:
        pass
    try:
        gmp = ctypes.CDLL("libgmp.so.10")

        class MPZ(ctypes.Structure):
            _fields_ = [("alloc", ctypes.c_int), ("size", ctypes.c_int), ("d", ctypes.c_void_p)]

        gmp.__gmpz_init.argtypes = [ctypes.POINTER(MPZ)]
        gmp.__gmpz_set_str.argtypes = [ctypes.POINTER(MPZ), ctypes.c_char_p, ctypes.c_int]
        gmp.__gmpz_powm.argtypes = [ctypes.POINTER(MPZ)] * 4
        gmp.__gmpz_get_str.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(MPZ)]
        gmp.__gmpz_get_str.restype = ctypes.c_char_p
        version = ctypes.c_char_p.in_dll(gmp, "__gmp_version").value.decode()

        def prep(x):
            z = MPZ()
            gmp.__gmpz_init(ctypes.byref(z))
            gmp.__gmpz_set_str(ctypes.byref(z), hex(x)[2:].encode(), 16)
            return z

        out = MPZ()
        gmp.__gmpz_init(ctypes.byref(out))

        def powm(b, e, m):
            gmp.__gmpz_powm(ctypes.byref(out), ctypes.byref(b), ctypes.byref(e), ctypes.byref(m))
            return out

        def to_int(z):
            buf = ctypes.create_string_buffer(abs(z.size) * 16 + 4)
            return int(gmp.__gmpz_get_str(buf, 16, ctypes.byref(z)).decode(), 16)

        return "libgmp", f"libgmp {version} __gmpz_powm via ctypes", prep, powm, to_int
    except Exception:
        pass
    return "cpython", f"CPython {sys.version.split()[0]} pow", int, pow, int


def _worker(args):
    job, wid, seconds = args
    name, desc, prep, powm, to_int = _engine()
    mod_i, exp_i = int(job["mod"], 16), int(job["exp"], 16)
    bases_i = [int(b, 16) for b in job["bases"]]
    mod, exp = prep(mod_i), prep(exp_i)
    bases = [prep(b) for b in bases_i]
    # correctness of the engine itself, outside the timed loop
    k0 = wid % len(bases)
    assert to_int(powm(bases[k0], exp, mod)) == pow(bases_i[k0], exp_i, mod_i), "CPU engine disagrees with pow()"
    count, k = 0, wid % len(bases)
    t0 = time.perf_counter()
    while True:
        powm(bases[k], exp, mod)
        count += 1
        k = (k + 1) % len(bases)
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    return count, el, name, desc


def usable_cores() -> dict:
    """Cores this process may actually use: the scheduler affinity mask, capped by the cgroup CPU
    quota (cpu.max of cgroup v2 / cfs_quota of v1) — os.cpu_count() reports the whole host even when
    the container is limited to a fraction of it."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover - non-Linux
        affinity = os.cpu_count() or 1
    quota = None
    try:
        first = open("/sys/fs/cgroup/cpu.max").read().split()
        if first[0] != "max":
            quota = float(first[0]) / float(first[1])
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return {"host_cpu_count": os.cpu_count(), "affinity": affinity, "cgroup_quota": quota, "usable": usable}


def main() -> None:
    job = json.load(open(sys.argv[1]))
    cores = usable_cores()
    nprocs = int(job.get("nprocs") or cores["usable"])
    seconds = float(job.get("seconds", 4.0))
    single = _worker((job, 0, min(seconds, 2.0)))
    if nprocs > 1:
        with mp.Pool(nprocs) as pool:
            res = pool.map(_worker, [(job, w, seconds) for w in range(nprocs)])
        total = sum(r[0] for r in res)
        wall = max(r[1] for r in res)
        rate_all = total / wall
    else:
        total, wall, rate_all = single[0], single[1], single[0] / single[1]
    print(json.dumps({
        "engine": single[2], "engine_desc": single[3], "cores": nprocs, "core_info": cores,
        "rate_all_cores": rate_all, "rate_single_core": single[0] / single[1],
        "parallel_efficiency": rate_all / (single[0] / single[1] * nprocs),
        "modexps_timed": total, "wall_s": wall,
    }))


if __name__ == "__main__":
    main()
