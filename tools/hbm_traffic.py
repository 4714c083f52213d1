#!/usr/bin/env python3
"""Measured HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot
share a pass on gfx950), written to profiles/<PROFILE_TAG, default r06>_hbm_traffic.json where bench.py picks it up by key.

usage: hbm_traffic.py <key> <kernel-name-substring> <fetch_pass_dir> <write_pass_dir> [<last N launches> [<dispatches per launch>]]

(dispatches per launch: a friendly-modulus one-wavefront exponentiation is two dispatches of powmod_n2_kernel — the tape
and, on the plain instance, its last product with the epilogue — whose traffic is summed.)

Corrections (MI355X_MICROARCH.md, HBM): the counters are in KiB; FETCH_SIZE tallies the 128-byte
requests of coalesced reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as is.  The access
pattern here is one dword per lane (256-byte rows per wavefront instruction), which the guide lists
as uncalibrated — bench.py therefore prints the modelled traffic (slot accesses of the tape x row
size) next to this figure, and the two agree within ~15 % (the model is an upper bound: reads served
by the L2 never reach the memory-side counters).
"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

import os

OUT = Path(__file__).resolve().parent.parent / "profiles" / (os.environ.get("PROFILE_TAG", "r06") + "_hbm_traffic.json")


def mean_counter(d, kernel_sub, counter, last, per_launch=1):
    vals = []
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and kernel_sub in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals += [float(r["Counter_Value"]) for r in rows]
    if last:
        vals = vals[-last * per_launch:]
    launches = len(vals) // per_launch
    return (sum(vals) / launches, launches) if launches else (None, 0)


def main():
    key, sub, fdir, wdir = sys.argv[1:5]
    last = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    dpl = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    fetch, nf = mean_counter(fdir, sub, "FETCH_SIZE", last, dpl)
    write, nw = mean_counter(wdir, sub, "WRITE_SIZE", last, dpl)
    assert fetch is not None and write is not None, "counter rows not found"
    data = json.loads(OUT.read_text()) if OUT.exists() else {}
    data[key] = {
        "bytes": (2 * fetch + write) * 1024,
        "fetch_size_kib_raw": fetch, "write_size_kib_raw": write, "dispatches": [nf, nw],
        "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of kernel '{sub}', mean per launch; "
                  "bytes = (2 x FETCH_SIZE + WRITE_SIZE) KiB (gfx950 correction for coalesced reads)",
    }
    OUT.write_text(json.dumps(data, indent=1, sort_keys=True) + "\n")
    print(key, data[key])


if __name__ == "__main__":
    main()
