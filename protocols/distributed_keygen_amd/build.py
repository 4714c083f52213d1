"""Builds libmxpaillier.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libmxpaillier.so"
SOURCES = [CSRC / "mx_capi.hip"]
HEADERS = sorted(CSRC.glob("*.hpp")) + [PKG.parent.parent / "include" / "mxpaillier.h"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any(p.stat().st_mtime > t for p in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the shared library for gfx950; returns its path."""
    if not force and not needs_build():
        return LIB
    cmd = [
        _hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
        "-Wno-unused-value", "-Wno-pass-failed",
        *[str(s) for s in SOURCES], "-o", str(LIB),
    ]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
