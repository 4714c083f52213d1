#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into one text summary for profiles/.

usage: prof_summary.py <out.txt> <stats_dir> [<pmc_dir> ...]
"""
import collections
import csv
import glob
import sys


def short(name: str) -> str:
    return name if len(name) <= 90 else name[:87] + "..."


def main() -> None:
    out, stats_dir, pmc_dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    lines = []
    for f in glob.glob(stats_dir + "/**/*_kernel_stats.csv", recursive=True):
        lines.append(f"## kernel stats (rocprofv3 --kernel-trace --stats): {f}")
        lines.append(f"{'kernel':92s} {'calls':>6s} {'avg_ms':>12s} {'min_ms':>12s} {'max_ms':>12s} {'pct':>7s}")
        for r in csv.DictReader(open(f)):
            lines.append(f"{short(r['Name']):92s} {r['Calls']:>6s} {float(r['AverageNs'])/1e6:12.4f} "
                         f"{float(r['MinNs'])/1e6:12.4f} {float(r['MaxNs'])/1e6:12.4f} {float(r['Percentage']):7.3f}")
    for d in pmc_dirs:
        for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            lines.append("")
            lines.append(f"## PMC pass (per-dispatch average): {f}")
            for k, v in agg.items():
                if not k.startswith("void mx::") and "mx::" not in k:
                    continue
                for c, x in sorted(v.items()):
                    lines.append(f"{short(k):92s} {c:24s} {sum(x)/len(x):20.1f}  (n={len(x)})")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
