"""Builds libmxpaillier.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc."""

from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libmxpaillier.so"
CODEC_SRC = CSRC / "mx_pycodec.c"
CODEC = PKG / "_mxcodec.so"          # CPython helper: bulk Python int <-> limb rows (host side, no arithmetic)
SOURCES = [CSRC / "mx_capi.hip", CSRC / "mx_capi_n2.hip", CSRC / "mx_capi_n2w.hip", CSRC / "mx_capi_n2s.hip",
           CSRC / "mx_capi_n2sw.hip"]
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
    if any(p.stat().st_mtime > t for p in SOURCES + HEADERS):
        return True
    return not CODEC.exists() or CODEC_SRC.stat().st_mtime > CODEC.stat().st_mtime


def build_codec(verbose: bool = False) -> Path:
    """gcc build of the CPython int <-> rows helper against the running interpreter's headers."""
    import sysconfig

    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("gcc not found")
    cmd = [cc, "-O2", "-shared", "-fPIC", "-I" + sysconfig.get_paths()["include"], str(CODEC_SRC), "-o", str(CODEC)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return CODEC


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the shared library for gfx950; returns its path."""
    if not force and not needs_build():
        return LIB
    build_codec(verbose)
    if not force and LIB.exists() and not any(p.stat().st_mtime > LIB.stat().st_mtime for p in SOURCES + HEADERS):
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    hipcc = _hipcc()
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed"]

    def compile_one(src: Path) -> Path:
        obj = objdir / (src.stem + ".o")
        cmd = [hipcc, *flags, "-c", str(src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=str(CSRC))
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:      # translation units in parallel
        objs = list(pool.map(compile_one, SOURCES))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *[str(o) for o in objs], "-o", str(LIB)]
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True, cwd=str(CSRC))
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
