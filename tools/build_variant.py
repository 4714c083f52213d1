#!/usr/bin/env python3
"""Builds libmxpaillier with extra compiler flags into protocols/distributed_keygen_amd/build/variants/<name>.so
(developer tool for A/B runs through MX_LIBRARY, tools/ab_variants.sh).  usage: build_variant.py <name> [flags ...]"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from protocols.distributed_keygen_amd import build as B
name, extra = sys.argv[1], sys.argv[2:]
out = B.PKG / "build" / "variants"; objdir = out / name; objdir.mkdir(parents=True, exist_ok=True)
flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", *extra]
def one(src):
    obj = objdir / (src.stem + ".o")
    subprocess.run([B._hipcc(), *flags, "-c", str(src), "-o", str(obj)], check=True, cwd=str(B.CSRC))
    return obj
with ThreadPoolExecutor(max_workers=len(B.SOURCES)) as pool:
    objs = list(pool.map(one, B.SOURCES))
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(out / f"{name}.so")], check=True)
for o in objs: os.remove(o)
print(out / f"{name}.so")
