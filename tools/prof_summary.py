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
    for f in glob.glob(stats_dir + "/**/*_kernel_trace.csv", recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mx::powmod" in r["Kernel_Name"]:
                per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
        for k, v in per.items():
            v.sort()
            d = [x[1] for x in v]
            lines.append("")
            lines.append(f"## launches of {short(k)} in start order (ms): " + " ".join(f"{x:.1f}" for x in d))
            for tail in (48, 12, 9, 6):
                if len(d) > tail:
                    lines.append(f"   mean of the last {tail} (bench.py's timed steps when --steps {tail}): {sum(d[-tail:]) / tail:.3f} ms")
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
