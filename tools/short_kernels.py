#!/usr/bin/env python3
"""Instruction counts and durations of the short kernels of the path (Jacobi filter, sieve, recombination, verdict, Shamir
interpolation) at the shapes bench.py's `short_kernels` leg times them — VERDICT r04 item 8.

  run                      launches every case of bench.short_kernel_cases REPS times (what the profiler wraps):
      cd /tmp; rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d <dir>/pmc -- python3 tools/short_kernels.py run
      cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d <dir>/trace -- python3 tools/short_kernels.py run
  fit <pmc dir> <trace dir> [out.json]
      joins the two passes: per case the mean SQ_INSTS_VALU and SQ_WAVES per launch (counter pass) and the mean duration
      (trace pass, unperturbed by counters) -> profiles/r05_short_kernels.json + a table on stdout.

Fractions are against the measured VALU issue rates (bench.py: 2.28 cycles per plain instruction and SIMD, 4.19 per integer
multiply); the mix of these kernels is not split by class, so both bounds are printed.
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
REPS = 6


def run() -> None:
    import bench
    import torch

    from protocols.distributed_keygen_amd import Engine

    eng = Engine(0)
    cases, keep = bench.short_kernel_cases(eng, torch)
    eng.set_priority_aux(False)
    torch.cuda.synchronize()
    for name, kernel, units, unit, fn in cases:
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()
        print(name, kernel, units, unit, flush=True)


def fit(pmc_dir: str, trace_dir: str, out_path: Path) -> None:
    import bench

    subs = {"jacobi": ("jacobi_kernel", "symbols", 4096 * 160), "sieve": ("sieve_kernel", "candidates", 65536),
            "combine": ("combine_kernel", "ciphertexts", 10000), "verdict": ("verdict_kernel", "slots", 4096 * 40),
            "lincomb": ("lincomb_kernel", "candidates", 65536)}
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pmc_dir + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            counters[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    durations = collections.defaultdict(list)
    for f in glob.glob(trace_dir + "/**/*_kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            durations[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
    out = {}
    print(f"{'kernel':10s} {'units/launch':>12s} {'ms':>8s} {'units/s':>12s} {'VALU instr/launch':>18s} {'waves':>8s} {'instr/unit':>11s} "
          f"{'frac (plain 2.28)':>18s} {'frac (all multiplies 4.19)':>27s}")
    for name, (sub, unit, units) in subs.items():
        ks = [k for k in counters if sub in k and "fallback" not in k]
        kd = [k for k in durations if sub in k and "fallback" not in k]
        if not ks or not kd:
            print(name, "NOT FOUND in the passes")
            continue
        # the timed launches are the last REPS dispatches of the case's kernel (set-up launches of other shapes come first)
        def last(seq):
            seq = sorted(seq)[-REPS:]
            return sum(v for _, v in seq) / len(seq)

        instr = sum(last(counters[k]["SQ_INSTS_VALU"]) for k in ks if counters[k]["SQ_INSTS_VALU"])
        waves = sum(last(counters[k]["SQ_WAVES"]) for k in ks if counters[k]["SQ_WAVES"])
        ms = sum(last(durations[k]) for k in kd)
        frac_plain = instr / (ms * 1e-3) / bench.PLAIN_ISSUE_PEAK
        frac_mul = instr / (ms * 1e-3) / bench.MAC_ISSUE_PEAK
        out[name] = {"kernel": sub, "units_per_launch": units, "unit": unit, "valu_instructions_per_launch": instr,
                     "waves_per_launch": waves, "kernel_ms": ms, "instructions_per_unit": instr / units,
                     "frac_vs_plain_valu_issue": frac_plain, "frac_if_all_multiplies": frac_mul,
                     "kernels_matched": ks,
                     "source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES (mean of the last launches) + a separate --kernel-trace pass for the duration"}
        print(f"{name:10s} {units:12d} {ms:8.3f} {units / ms * 1e3:12.4g} {instr:18.4g} {waves:8.0f} {instr / units:11.1f} {frac_plain:18.3f} {frac_mul:27.3f}")
    out_path.write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print("wrote", out_path)


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run()
    else:
        fit(sys.argv[2], sys.argv[3], Path(sys.argv[4]) if len(sys.argv) > 4 else ROOT / "profiles" / "r05_short_kernels.json")
