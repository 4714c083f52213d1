#!/usr/bin/env python3
"""Builds libmxpaillier with extra compiler flags and/or another run length of the alignment pass into
protocols/distributed_keygen_amd/build/variants/<name>.so (developer tool for A/B runs through MX_LIBRARY,
tools/ab_variants.sh, tools/variant_probe.py).
usage: build_variant.py <name> [--align-run X] [flags ...]      (X: cost of an s_nop in the alignment pass, 0 = no pass)"""
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from protocols.distributed_keygen_amd import build as B
args = sys.argv[1:]
name = args.pop(0)
align_run = None
if args and args[0] == "--align-run":
    align_run = float(args[1]); args = args[2:]
out = B.PKG / "build" / "variants"; objdir = out / name; objdir.mkdir(parents=True, exist_ok=True)
with ThreadPoolExecutor(max_workers=len(B.SOURCES)) as pool:
    objs = list(pool.map(lambda src: B.compile_unit(src, objdir, args, align_run), B.SOURCES))
print(B.link(objs, out / f"{name}.so"))
for o in objs:
    o.unlink()
objdir.rmdir()
