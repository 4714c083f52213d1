#!/usr/bin/env python3
"""Instruction mix of the hot basic blocks of a kernel in a hipcc -save-temps .s file.

usage: isa_mix.py file.s kernel_substring [min_mads]
"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
want = sys.argv[2]
min_mads = int(sys.argv[3]) if len(sys.argv) > 3 else 50
for m in re.finditer(r'^(\S+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    parts = re.split(r'^(\.LBB[0-9_]+):.*$', body, flags=re.M)
    total = 0
    print("kernel", name)
    for i in range(0, len(parts)):
        if i % 2 == 1:
            continue
        label = parts[i - 1] if i > 0 else "entry"
        ins = [l.split()[0] for l in (x.strip() for x in parts[i].split('\n')) if l and l[0] not in '.;/']
        total += len(ins)
        c = collections.Counter(ins)
        if c.get('v_mad_u64_u32', 0) >= min_mads:
            print(f"  {label}: {len(ins)} instrs, {c['v_mad_u64_u32']} v_mad_u64_u32 ({100.0*c['v_mad_u64_u32']/len(ins):.1f}%)")
            print("   ", ", ".join(f"{k}:{v}" for k, v in c.most_common(30)))
    print("  total instructions:", total)
