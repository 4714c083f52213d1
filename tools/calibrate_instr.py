#!/usr/bin/env python3
"""VALU instruction-count model of the modexp kernels, calibrated against SQ_INSTS_VALU.

bench.py's primary roofline is VALU instruction issue.  The number of VALU wave-instructions a
launch executes is a function of the kernel instance and of the exponent only (control flow is
uniform, there is no data-dependent work besides a few carry-ripple iterations):

    per wavefront:  n_sqr * I_sqr + n_mul * I_mul + F        with (I_sqr, I_mul, F) per (L, nblk)

where n_sqr / n_mul are the squarings / multiplications of the exponentiation (for the N^2 pair
kernel the plan reports them: mx_nsquare_plan.n_sqr / .n_mul; for the fixed-window generic kernel
they follow from the window width and the digit count).  This tool measures the three constants
for every (L, nblk):

  python3 tools/calibrate_instr.py run   <configs.json>        launches 4 exponents per (kernel, L, nblk)
        — run it under  rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d <dir>
  python3 tools/calibrate_instr.py fit   <configs.json> <dir> <model.json>
        — matches the dispatches to the configs in launch order, solves the 3x3 system from the first
          three exponents and reports the residual of the held-out fourth (a check of the linear model)

The model is committed as profiles/r0N_instr_model.json together with a digest of the MACHINE CODE of the modelled
kernels in the library it was fitted to (kernel_code_digest; models of rounds 2-5 recorded a digest of the source headers);
bench.py reads it and refuses it (roofline fraction null, with the reason) when the digest no longer matches the library
it runs.  Kinds: "n2" (pair kernel, one wavefront per group), "n2split" (two
wavefronts per group: the constants are per PAIR of wavefronts), "generic" (fixed-window kernel).
"""

from __future__ import annotations

import csv
import glob
import json
import random
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

W = 29
WAVES = 4
KERNEL_SOURCES = ["mx_lanes.hpp", "mx_mont.hpp", "mx_powmod.hpp", "mx_powmod_n2.hpp", "mx_powmod_n2_split.hpp"]     # the sources of the MODELLED kernels (the bipartite form, mx_bimont.hpp, has no entry)


def kernel_sources_digest() -> str:
    """sha256 over the device-code headers whose instruction streams the model describes."""
    import hashlib

    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        h.update((ROOT / "protocols" / "distributed_keygen_amd" / "csrc" / name).read_bytes())
    return h.hexdigest()



MODELLED_KERNELS = ("powmod_kernel", "powmod_n2_kernel", "powmod_n2_split_kernel")


def kernel_code_digest(lib_path=None) -> str:
    """sha256 over the MACHINE CODE of the modelled kernels in the built library (every instance of the three modexp kernel
    templates: the bytes of its function symbol in the gfx950 code objects, by mangled name) — what the constants of
    the model were actually fitted to.  A comment, a renamed macro or a new kernel elsewhere in the library leaves it
    alone; any change of the instruction stream of a modelled kernel changes it.  (The digest over the source headers,
    kernel_sources_digest, refused the model after every edit of those files, including the ones that generate the same
    code.)"""
    import hashlib
    import struct

    from tools import scratch_report

    if lib_path is None:
        from protocols.distributed_keygen_amd import _lib

        lib_path = _lib.LIB_PATH
    funcs = {}
    for elf in scratch_report.code_objects_of_library(lib_path):
        shoff, = struct.unpack_from("<Q", elf, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
        secs = []
        for i in range(shnum):
            sh = shoff + i * shentsize
            sh_type, = struct.unpack_from("<I", elf, sh + 4)
            addr, off, size, link = struct.unpack_from("<QQQI", elf, sh + 0x10)
            secs.append((sh_type, addr, off, size, link))
        for sh_type, _addr, off, size, link in secs:
            if sh_type != 2:                 # SHT_SYMTAB
                continue
            str_off = secs[link][2]
            for p in range(off, off + size, 24):
                st_name, st_info, _other, shndx, value, st_size = struct.unpack_from("<IBBHQQ", elf, p)
                if st_info & 0xF != 2 or not st_size or shndx == 0 or shndx >= len(secs):      # STT_FUNC, defined
                    continue
                end = elf.index(b"\0", str_off + st_name)
                name = elf[str_off + st_name:end].decode()
                if not any(f"2mx{len(k)}{k}I" in name for k in MODELLED_KERNELS):
                    continue
                _t, s_addr, s_off, _s, _l = secs[shndx]
                start = s_off + (value - s_addr)
                funcs[name] = elf[start:start + st_size]
    if len(funcs) < 60:
        raise RuntimeError(f"only {len(funcs)} modexp kernel symbols found in {lib_path}")
    h = hashlib.sha256()
    for name in sorted(funcs):
        h.update(name.encode() + b"\0" + struct.pack("<Q", len(funcs[name])) + funcs[name])
    return h.hexdigest()


def fixed_window(exp_bits: int) -> int:
    """mx_host.hpp: fixed_window (minimise ceil(bits/w) multiplications + 2^w - 2 table products)."""
    best, bestc = 1, None
    for w in range(1, 8):
        c = (exp_bits + w - 1) // w + (1 << w) - 2
        if bestc is None or c < bestc:
            best, bestc = w, c
    return best


def generic_counts(max_ebits: int, exp_limbs: int):
    """(n_sqr, n_mul) of powmod_kernel<.., false>: window squarings and table + window multiplications."""
    win = fixed_window(32 * exp_limbs)
    ndigits = max(1, -(-max_ebits // win))
    return (ndigits - 1) * win, ((1 << win) - 2) + (ndigits - 1)


def exponents(rng: random.Random, kind: str):
    """Four exponents per instance: three to fit (I_sqr, I_mul, F), the last one held out.
    n2 (sliding window): the multiplication count follows the exponent's density, the squaring count
    its length.  generic (fixed window): both follow the length only (every digit multiplies), and
    the window width changes with it — three lengths with three different widths."""
    def rnd(bits):
        return rng.getrandbits(bits) | (1 << (bits - 1)) | 1

    if kind in ("n2", "n2split"):
        return [(1 << 191, 0), ((1 << 192) - 1, 0), (rnd(120), 0), (rnd(192), 0)]
    # (exponent, exponent row width in words): the row width selects the window (mx_host.hpp fixed_window),
    # so a short exponent in wide rows has few squarings and a large table — what separates I_mul from I_sqr
    return [(rnd(60), 65), (rnd(60), 2), (rnd(600), 19), (rnd(192), 6)]


def run(cfg_path: str) -> None:
    import torch  # noqa: F401

    from protocols.distributed_keygen_amd import Engine, limbs as Lm

    eng = Engine()
    rng = random.Random(20260201)
    configs = []
    for kind in ("n2", "n2split", "generic"):
        for L in ((9, 18) if kind == "n2" else (3, 9, 18)):
            max_nblk = {("n2", 9): 32, ("n2", 18): 16, ("generic", 3): 64, ("generic", 9): 64, ("generic", 18): 32,
                        ("n2split", 3): 64, ("n2split", 9): 32, ("n2split", 18): 16}[(kind, L)]
            eng.set_wavefronts_per_group(2 if kind == "n2split" else 1)
            for nblk in range(1, max_nblk + 1):
                bits = W * L * nblk - (4 if L != 3 else 4 + W + 2)          # the head room of mx_host.hpp: choose_geometry
                if kind == "n2split" and L == 9 and 4 < nblk <= 16:
                    bits = W * L * nblk - (4 + W + 2)      # groups of 8 / 16 lanes: the friendly instances (what key_length 2048 / 4096 run)
                friendly_1w = kind == "n2" and L == 18 and 2 < nblk <= 8
                if friendly_1w:
                    bits = W * L * nblk - (4 + W + 2)      # groups of 4 / 8 lanes: the friendly one-wavefront instances (round 4)
                if bits < 8:
                    continue
                k = 1
                while k < nblk:
                    k *= 2
                batch = WAVES * (64 // k)
                eng.set_limbs_per_lane(L)
                n = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
                for e, erow in exponents(rng, kind):
                    if kind in ("n2", "n2split"):
                        assert eng.nsquare_launch_shape(bits, batch) == (k, L, W, nblk, 2 if kind == "n2split" else 1)
                        n2 = n * n
                        rows = eng.to_device(Lm.pack([rng.randrange(n2) for _ in range(batch)], Lm.limbs_for(n2)))
                        plan = eng.nsquare_plan(n, e)
                        eng.powmod_nsquare_t(rows, n, e, segments=1)
                        # a friendly one-wavefront exponentiation is two dispatches of powmod_n2_kernel: the tape on the
                        # friendly instance, its last product and the epilogue on the plain one
                        configs.append({"kind": kind, "L": L, "nblk": nblk, "K": k, "waves": WAVES, "dispatches": 2 if friendly_1w else 1,
                                        "n_sqr": int(plan.desc.n_sqr), "n_mul": int(plan.desc.n_mul)})
                    else:
                        assert eng.geometry(bits, batch * 2, 2) == (k, L, W, nblk)
                        groups = 2
                        mods = [n, n - 2]
                        rows = eng.to_device(Lm.pack([rng.randrange(n - 2) for _ in range(batch * groups)], Lm.limbs_for(n)))
                        exps = (eng.to_device(Lm.pack([e, e - 2], erow)), e.bit_length())
                        eng.powmod_multi_t(rows, mods, exps, batch)
                        nsq, nmu = generic_counts(e.bit_length(), erow)
                        configs.append({"kind": kind, "L": L, "nblk": nblk, "K": k, "waves": WAVES * groups,
                                        "n_sqr": nsq, "n_mul": nmu})
                eng.synchronize()
    Path(cfg_path).write_text(json.dumps(configs))
    print(f"{len(configs)} launches")


def solve3(rows, rhs):
    """Gaussian elimination of a 3x3 system."""
    a = [list(map(float, r)) + [float(b)] for r, b in zip(rows, rhs)]
    for i in range(3):
        piv = max(range(i, 3), key=lambda r: abs(a[r][i]))
        a[i], a[piv] = a[piv], a[i]
        for r in range(3):
            if r != i:
                f = a[r][i] / a[i][i]
                a[r] = [x - f * y for x, y in zip(a[r], a[i])]
    return [a[i][3] / a[i][i] for i in range(3)]


def fit(cfg_path: str, pmc_dir: str, model_path: str) -> None:
    configs = json.loads(Path(cfg_path).read_text())
    disp = {"n2": [], "n2split": [], "generic": []}
    for f in glob.glob(pmc_dir + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != "SQ_INSTS_VALU":
                continue
            name = r["Kernel_Name"]
            kind = ("n2" if "powmod_n2_kernel" in name else "n2split" if "powmod_n2_split_kernel" in name
                    else "generic" if "mx::powmod_kernel" in name else None)
            if kind:
                disp[kind].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]), name))
    model = {"n2": {"9": {}, "18": {}}, "n2split": {"3": {}, "9": {}, "18": {}}, "generic": {"3": {}, "9": {}, "18": {}}, "max_residual": 0.0,
             "kernel_code_sha256": kernel_code_digest(), "kernel_code_of": list(MODELLED_KERNELS),
             "kernel_sources_sha256": kernel_sources_digest(), "kernel_sources": KERNEL_SOURCES,
             "source": "tools/calibrate_instr.py: SQ_INSTS_VALU of 4 exponents per (kernel, L, nblk); wave-instructions "
                       "per wavefront (n2split: per pair of wavefronts) = n_sqr*I_sqr + n_mul*I_mul + F"}
    for kind in ("n2", "n2split", "generic"):
        d = sorted(disp[kind])
        cfgs = [c for c in configs if c["kind"] == kind]
        assert len(d) == sum(c.get("dispatches", 1) for c in cfgs), (kind, len(d), len(cfgs))
        totals, pos = [], 0
        for c in cfgs:                                    # the dispatches of one exponentiation, summed
            nd = c.get("dispatches", 1)
            totals.append(sum(x[1] for x in d[pos : pos + nd]))
            pos += nd
        for i in range(0, len(cfgs), 4):
            grp = cfgs[i : i + 4]
            per_wave = [totals[i + j] / grp[j]["waves"] for j in range(4)]
            fit_rows, held = (0, 1, 2), 3
            sol = solve3([[grp[j]["n_sqr"], grp[j]["n_mul"], 1] for j in fit_rows], [per_wave[j] for j in fit_rows])
            pred = sol[0] * grp[held]["n_sqr"] + sol[1] * grp[held]["n_mul"] + sol[2]
            resid = abs(pred - per_wave[held]) / per_wave[held]
            model["max_residual"] = max(model["max_residual"], resid)
            model[kind][str(grp[0]["L"])][str(grp[0]["nblk"])] = [round(sol[0], 2), round(sol[1], 2), round(sol[2], 1)]
    Path(model_path).write_text(json.dumps(model, indent=0))
    print("max residual of the held-out exponent:", model["max_residual"])
    for kind in ("n2", "n2split", "generic"):
        for L in ("3", "9", "18"):
            for nblk in ("4", "8", "16", "24"):
                if L not in model[kind]:
                    continue
                if nblk in model[kind][L]:
                    print(kind, "L", L, "nblk", nblk, model[kind][L][nblk])


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        fit(sys.argv[2], sys.argv[3], sys.argv[4])
