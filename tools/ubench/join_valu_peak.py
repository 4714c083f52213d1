#!/usr/bin/env python3
"""Joins the plain run of tools/ubench/valu_peak (wall clock, in-kernel clock) with the rocprofv3 --pmc passes of
the same binary by dispatch order.  usage: join_valu_peak.py <dir with valu_peak.txt, pmc_a_counters.csv,
pmc_b_counters.csv> > profiles/r03_ubench_valu_peak.txt"""
import csv
import re
import sys
from collections import defaultdict

d = sys.argv[1]
rows = {}
for line in open(f"{d}/valu_peak.txt"):
    m = re.match(r"dispatch\s+(\d+)\s+(.+?)\s+waves/SIMD=(\d) simds/CU=(\d)\s+active_waves=(\d+).*wall=([\d.]+) ms\s+clock=(\d+) MHz\s+"
                 r"cycles/instr/SIMD: ([\d.]+) @measured clock, ([\d.]+) @2400", line)
    if m:
        rows[int(m.group(1)) + 1] = dict(kind=m.group(2), w=int(m.group(3)), s=int(m.group(4)), waves=int(m.group(5)),
                                         wall=float(m.group(6)), clk=int(m.group(7)), cyc=float(m.group(8)), cyc24=float(m.group(9)))
    elif line.startswith("device"):
        header = line.strip()
ctr = defaultdict(dict)
for f in ("pmc_a_counters.csv", "pmc_b_counters.csv"):
    for r in csv.DictReader(open(f"{d}/{f}")):
        did = int(r["Dispatch_Id"])
        ctr[did][r["Counter_Name"]] = float(r["Counter_Value"])
        ctr[did]["dur_" + f[:5]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
print(header)
print("VALU issue cost on gfx950, 16 independent instructions per stream, 256 000 instructions per wavefront, one dispatch per row.")
print("cyc = wall time x in-kernel clock / (instructions x wavefronts per populated SIMD); clk = s_memtime / s_memrealtime x 100 MHz;")
print("clk_grbm = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration of the counter pass; valu_instr = SQ_INSTS_VALU (wave-instructions);")
print("act/instr = SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU (1.00 in every row: on gfx950 this counter advances once per VALU instruction, it does")
print("not measure issue cycles; the issue cost is the wall-clock column).")
print(f"{'instruction':31s} {'w/SIMD':>6s} {'SIMDs':>5s} {'wall ms':>8s} {'clk MHz':>8s} {'clk_grbm':>8s} {'cyc@clk':>8s} {'cyc@2.4G':>8s} "
      f"{'valu_instr':>12s} {'expected':>12s} {'act/instr':>10s}")
for did in sorted(rows):
    r, c = rows[did], ctr.get(did, {})
    grbm = c.get("GRBM_GUI_ACTIVE", 0) / 8 / (c.get("dur_pmc_b", 1) * 1e-3) / 1e6 if c.get("GRBM_GUI_ACTIVE") else float("nan")
    insts = c.get("SQ_INSTS_VALU", float("nan"))
    busy = c.get("SQ_ACTIVE_INST_VALU", float("nan")) / insts if insts == insts and insts else float("nan")
    print(f"{r['kind']:31s} {r['w']:6d} {r['s']:5d} {r['wall']:8.3f} {r['clk']:8d} {grbm:8.0f} {r['cyc']:8.3f} {r['cyc24']:8.3f} "
          f"{insts:12.4g} {r['waves'] * 256000:12.4g} {busy:10.2f}")
