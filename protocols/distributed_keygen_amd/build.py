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
           CSRC / "mx_capi_n2sw.hip", CSRC / "mx_capi_lat.hip", CSRC / "mx_capi_bip.hip"]
HEADERS = sorted(CSRC.glob("*.hpp")) + [PKG.parent.parent / "include" / "mxpaillier.h"]
BUILD_INPUTS = [Path(__file__).resolve(), PKG / "asm_align.py"]      # the build recipe itself


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


STAMP = PKG / "build" / "sources.sha256"      # digest of everything the library was built from, written after the link


def sources_digest() -> str:
    import hashlib

    h = hashlib.sha256()
    for p in SOURCES + HEADERS + BUILD_INPUTS:
        h.update(p.name.encode() + b"\0" + p.read_bytes() + b"\0")
    return h.hexdigest()


def _lib_stale() -> bool:
    """The library against the sources it was built from: by content where the build left its stamp (a checkout or a
    copy of the tree changes modification times without changing a byte), by modification time otherwise."""
    if not LIB.exists():
        return True
    try:
        return STAMP.read_text().strip() != sources_digest()
    except OSError:
        t = LIB.stat().st_mtime
        return any(p.stat().st_mtime > t for p in SOURCES + HEADERS + BUILD_INPUTS)


def needs_build() -> bool:
    if _lib_stale():
        return True
    return not CODEC.exists() or CODEC_SRC.stat().st_mtime > CODEC.stat().st_mtime


def build_codec(verbose: bool = False) -> Path:
    """gcc build of the CPython int <-> rows helper against the running interpreter's headers."""
    import sysconfig

    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise RuntimeError("gcc not found")
    cmd = [cc, "-O2", "-shared", "-fPIC", "-pthread", "-I" + sysconfig.get_paths()["include"], str(CODEC_SRC), "-o", str(CODEC)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return CODEC


FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-pass-failed", "-Wno-unused-command-line-argument"]
def compile_unit(src: Path, objdir: Path, extra_flags=(), align_run=None, verbose: bool = False) -> Path:
    """One translation unit -> host object with the device code embedded.  The device side goes through its assembly:
    hipcc -S, the alignment pass of asm_align.py (64-bit instructions on 8-byte addresses), assembler, lld, offload
    bundle; the host side is compiled with that bundle as its GPU binary — the steps `hipcc -c` runs internally,
    with one pass over the assembly in between.  align_run = 0 skips the pass (plain `hipcc -c`)."""
    from . import asm_align

    hipcc = _hipcc()
    run = asm_align.NOP_COST if align_run is None else align_run
    obj = objdir / (src.stem + ".o")
    flags = [*FLAGS, *extra_flags]

    def sh(cmd):
        if verbose:
            print(" ".join(map(str, cmd)))
        subprocess.run(list(map(str, cmd)), check=True, cwd=str(CSRC))

    if run <= 0:
        sh([hipcc, *flags, "-c", src, "-o", obj])
        return obj
    stem = objdir / src.stem
    asm, aligned, dev_obj, hsaco, fatbin = (Path(f"{stem}{ext}") for ext in (".s", ".aligned.s", ".dev.o", ".hsaco", ".hipfb"))
    sh([hipcc, *flags, "--cuda-device-only", "-S", src, "-o", asm])
    stats = asm_align.align_file(asm, aligned, dev_obj, run)
    if verbose:
        print(f"{src.name}: {sum(v for v in stats.values() if v > 0)} re-encodings / s_nop in {len(stats)} functions"
              + (f", {sum(1 for v in stats.values() if v == -2)} skipped" if any(v == -2 for v in stats.values()) else "")
              + (f", {sum(1 for v in stats.values() if v == -1)} NOT MATCHED" if any(v == -1 for v in stats.values()) else ""))
    asm_align.assemble(aligned, dev_obj)
    llvm = asm_align.LLVM_BIN
    sh([llvm / "lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev_obj])
    sh([llvm / "clang-offload-bundler", "-type=o", "-bundle-align=4096",
        "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", f"-input={hsaco}", f"-output={fatbin}"])
    sh([hipcc, *flags, "--cuda-host-only", "-c", src, "-o", obj, "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fatbin])
    for tmp in (asm, aligned, dev_obj, hsaco, fatbin):
        tmp.unlink(missing_ok=True)
    return obj


def link(objs, out: Path, verbose: bool = False) -> Path:
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *[str(o) for o in objs], "-o", str(out)]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=str(CSRC))
    return out


def build(force: bool = False, verbose: bool = False) -> Path:
    """Compile the shared library for gfx950; returns its path.  MX_BUILD_ALIGN_RUN in the environment of the BUILD
    (not read by the library): the cost of an inserted s_nop in the assembly alignment pass, 0 = no pass — for A/B runs
    of the pass itself."""
    if not force and not needs_build():
        return LIB
    build_codec(verbose)
    if not force and not _lib_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)
    align_run = float(os.environ["MX_BUILD_ALIGN_RUN"]) if os.environ.get("MX_BUILD_ALIGN_RUN") else None
    if align_run is None or align_run > 0:
        from . import asm_align

        note = asm_align.toolchain_note()
        if note:
            print(note)
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:      # translation units in parallel
        objs = list(pool.map(lambda src: compile_unit(src, objdir, (), align_run, verbose), SOURCES))
    digest = sources_digest()
    out = link(objs, LIB, verbose)
    STAMP.write_text(digest + "\n")
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
