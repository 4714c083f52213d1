#!/usr/bin/env python3
"""Per-kernel private-segment (scratch) sizes, spill counts and register counts of the library's device code.

  python tools/scratch_report.py [--all] [file.s | library.so ...]

Without files: compiles every translation unit of build.SOURCES to device assembly (hipcc --cuda-device-only -S, the
same flags as the build) into a temporary directory and reads the kernel descriptors' metadata.  Prints the kernels
whose private segment is not empty (all kernels with --all).  tests/test_instances.py asserts on the same data that no
modexp kernel uses scratch."""
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def kernels_of(asm_text):
    """[(mangled name, private_segment_fixed_size, vgpr_spill_count, sgpr_spill_count, vgpr_count)] from the
    amdhsa.kernels metadata of one assembly file."""
    out = []
    for blk in re.split(r"\n  - \.agpr_count:", asm_text)[1:]:
        def field(key, default="0"):
            m = re.search(r"^\s+\.%s:\s+(\S+)$" % key, blk, re.M)
            return m.group(1) if m else default
        name = field("name", "?")
        out.append((name, int(field("private_segment_fixed_size")), int(field("vgpr_spill_count")), int(field("sgpr_spill_count")),
                    int(field("vgpr_count"))))
    return out


def kernels_of_library(path):
    """The same tuples read from a BUILT library: every gfx950 code object of its offload bundles (.hip_fatbin), the
    msgpack metadata of the NT_AMDGPU_METADATA note — what actually ships, on any box, without a compiler."""
    import struct

    import msgpack

    data = Path(path).read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    pos = data.find(magic)
    while pos >= 0:
        (count,) = struct.unpack_from("<Q", data, pos + len(magic))
        cur = pos + len(magic) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", data, cur)
            triple = data[cur + 24 : cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "amdgcn" in triple and size:
                out.extend(_kernels_of_code_object(data[pos + off : pos + off + size], msgpack))
        pos = data.find(magic, pos + len(magic))
    return out


def code_objects_of_library(path):
    """The raw gfx950 code objects (ELF images) inside a built library's offload bundles."""
    import struct

    data = Path(path).read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = data.find(magic)
    while pos >= 0:
        (count,) = struct.unpack_from("<Q", data, pos + len(magic))
        cur = pos + len(magic) + 8
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", data, cur)
            triple = data[cur + 24 : cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "amdgcn" in triple and size:
                yield data[pos + off : pos + off + size]
        pos = data.find(magic, pos + len(magic))


def _kernels_of_code_object(elf: bytes, msgpack):
    import struct

    assert elf[:4] == b"\x7fELF" and elf[4] == 2, "64-bit ELF code object expected"
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    out = []
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type, = struct.unpack_from("<I", elf, sh + 4)
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        if sh_type != 7:            # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            desc = p + 12 + ((namesz + 3) & ~3)
            if ntype == 32 and elf[p + 12 : p + 12 + 6] == b"AMDGPU":
                meta = msgpack.unpackb(elf[desc : desc + descsz], raw=False, strict_map_key=False)
                for k in meta.get("amdhsa.kernels", []):
                    out.append((k[".name"], int(k.get(".private_segment_fixed_size", 0)), int(k.get(".vgpr_spill_count", 0)),
                                int(k.get(".sgpr_spill_count", 0)), int(k.get(".vgpr_count", 0))))
            p = desc + ((descsz + 3) & ~3)
    return out


def compile_all(outdir: Path):
    from protocols.distributed_keygen_amd import build as B

    def one(src):
        dst = outdir / (src.stem + ".s")
        subprocess.run([B._hipcc(), *B.FLAGS, "--cuda-device-only", "-S", str(src), "-o", str(dst)], check=True, cwd=str(B.CSRC),
                       stderr=subprocess.DEVNULL)
        return dst

    with ThreadPoolExecutor(max_workers=4) as pool:
        return list(pool.map(one, B.SOURCES))


def report(files, show_all=False):
    rows = []
    for f in files:
        if str(f).endswith(".so"):
            rows.extend((Path(f).stem,) + k for k in kernels_of_library(f))
            continue
        for k in kernels_of(Path(f).read_text()):
            rows.append((Path(f).stem,) + k)
    names = demangle([r[1] for r in rows])
    bad = 0
    for unit, name, priv, vs, ss, vg in rows:
        if priv or show_all:
            print(f"{unit:14s} private {priv:5d} B  vgpr spills {vs:3d}  sgpr spills {ss:3d}  vgprs {vg:3d}  {names[name][:110]}")
        bad += 1 if priv else 0
    print(f"{len(rows)} kernels, {bad} with a private segment")
    return rows


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    if args:
        report(args, "--all" in sys.argv)
    else:
        with tempfile.TemporaryDirectory() as tmp:
            report(compile_all(Path(tmp)), "--all" in sys.argv)
