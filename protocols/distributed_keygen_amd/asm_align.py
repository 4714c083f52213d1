"""Build step: 8-byte alignment of 64-bit instructions in the device assembly of the HIP kernels.

Measured on MI355X (profiles/r03_build_variants_ab.txt): a 64-bit instruction (VOP3 multiply-adds,
DPP moves — nearly everything in the Montgomery inner loops) that starts at an address = 4 mod 8 costs the
wavefront about 0.4 of an issue slot more than one that starts at 0 mod 8, and a run of them keeps its parity until
the next 32-bit instruction (s_waitcnt, s_nop, VOP2 adds).  Whether the long runs of a loop are aligned is decided
by how many 32-bit instructions happen to precede them — identical inner loops ran 122 or 140 ms depending on
unrelated code in front (the <8,18> two-wavefront instance of the pair kernel).  The compiler has no pass for this;
this one works on its assembly output:

    hipcc --cuda-device-only -S   ->  x.s
    assemble, llvm-objdump -d     ->  the size of every instruction, in order
    insert `s_nop 0` in front of every run of >= RUN 64-bit instructions that would start at 4 mod 8
    assemble the result (build.py links, bundles and embeds it into the host object)

An s_nop costs an issue slot itself, hence the minimum run length.  The pass changes no instruction and no
register; it only adds s_nop 0 between instructions — never inside an inline-asm block, and never within the
four instructions behind an s_getpc_b64: the s_add_u32 / s_addc_u32 that follow it carry sym@rel32 literals whose
addends (+4, +12) assume that they follow it directly.  If the assembly and the disassembly of a function cannot be
matched line by line the function is left as the compiler wrote it.
"""

from __future__ import annotations

import re
import subprocess
from pathlib import Path
from typing import Dict, List, Optional, Tuple

LLVM_BIN = Path("/opt/rocm/lib/llvm/bin")
RUN = 5           # minimum number of consecutive 64-bit instructions worth an s_nop (3, 5, 8 measured alike; 5 inserts a third fewer)
# Left alone: the instances with 3 limbs per lane (template arguments <K, 3, 29, ...>).  Their blocks are a few dozen
# instructions on ONE dependent chain (the latency geometry), where an s_nop costs more than the fetch it saves:
# 14.8 -> 15.3 ms and 55.9 -> 59.5 ms per decrypt with the pass, measured.
SKIP = re.compile(r"ELi3ELi29E")

_label = re.compile(r"^([A-Za-z_$][\w$.]*):")
_local_label = re.compile(r"^(\.L[\w$.]*):")
_dis_func = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
_dis_insn = re.compile(r"^\s+(\S+).*?//\s*[0-9A-F]+:\s*((?:[0-9A-F]{8}\s?)+)")


def _is_insn(line: str) -> bool:
    s = line.strip()
    if not s or s[0] in ".;#" or s.startswith("//") or s.endswith(":"):
        return False
    return True


def disassembly_sizes(obj: Path) -> Dict[str, List[Tuple[str, int]]]:
    """{function: [(mnemonic, bytes), ...]} from llvm-objdump -d."""
    out = subprocess.run([str(LLVM_BIN / "llvm-objdump"), "-d", str(obj)], check=True, capture_output=True, text=True).stdout
    funcs: Dict[str, List[Tuple[str, int]]] = {}
    cur: Optional[List[Tuple[str, int]]] = None
    for line in out.split("\n"):
        m = _dis_func.match(line)
        if m:
            cur = funcs.setdefault(m.group(1), [])
            continue
        if cur is None:
            continue
        m = _dis_insn.match(line)
        if m:
            cur.append((m.group(1), 4 * len(m.group(2).split())))
    return funcs


def align_text(asm: str, sizes: Dict[str, List[Tuple[str, int]]], run: int = RUN, skip=SKIP) -> Tuple[str, Dict[str, int]]:
    """Returns the assembly with s_nop 0 inserted, and {function: nops inserted} (-1: could not be matched, -2: skipped)."""
    lines = asm.split("\n")
    out: List[str] = []
    stats: Dict[str, int] = {}
    i = 0
    n = len(lines)
    while i < n:
        m = _label.match(lines[i])
        if not m or m.group(1).startswith(".L") or m.group(1) not in sizes:
            out.append(lines[i])
            i += 1
            continue
        name = m.group(1)
        if skip is not None and skip.search(name):
            stats[name] = -2
            out.append(lines[i])
            i += 1
            continue
        # the function's lines: up to its .Lfunc_end label
        j = i + 1
        while j < n and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i + 1 : j]
        dis = sizes[name]
        # match instruction lines with the disassembly (alignment padding shows up there as extra s_nop)
        seq: List[Tuple[int, int]] = []          # (index into body, size)
        k = 0
        ok = True
        for bi, line in enumerate(body):
            if not _is_insn(line):
                continue
            mnem = line.split()[0]
            while k < len(dis) and dis[k][0] != mnem and dis[k][0] in ("s_nop", "s_code_end"):
                k += 1
            if k >= len(dis) or dis[k][0] != mnem:
                ok = False
                break
            seq.append((bi, dis[k][1]))
            k += 1
        if not ok:
            stats[name] = -1
            out.extend(lines[i:j])
            i = j
            continue
        size_at = dict(seq)
        insn_idx = [bi for bi, _ in seq]
        pos_of = {bi: p for p, bi in enumerate(insn_idx)}
        inserted = 0
        offset = 0                                   # the function starts 256-byte aligned
        new_body: List[str] = []
        in_asm_block = False
        after_getpc = 0          # s_getpc_b64 + s_add_u32/s_addc_u32 with sym@rel32 literals: position dependent
        for bi, line in enumerate(body):
            s = line.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm_block = True
            elif s.startswith(";;#ASMEND"):
                in_asm_block = False
            if s.startswith(".p2align"):
                a = 1 << int(re.split(r"[\s,]+", s)[1])
                offset = (offset + a - 1) // a * a
            if bi in size_at:
                sz = size_at[bi]
                if sz % 8 == 0 and offset % 8 == 4 and not in_asm_block and after_getpc == 0:
                    # the run of 64-bit instructions that starts here
                    p = pos_of[bi]
                    length = 0
                    while p + length < len(seq) and seq[p + length][1] % 8 == 0:
                        length += 1
                    if length >= run:
                        new_body.append("\ts_nop 0")
                        offset += 4
                        inserted += 1
                offset += sz
                after_getpc = 4 if s.startswith("s_getpc_b64") else max(0, after_getpc - 1)
            new_body.append(line)
        stats[name] = inserted
        out.append(lines[i])
        out.extend(new_body)
        i = j
    return "\n".join(out), stats


def assemble(src: Path, obj: Path) -> None:
    subprocess.run([str(LLVM_BIN / "clang"), "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", str(src), "-o", str(obj)],
                   check=True)


def misaligned(sizes: Dict[str, List[Tuple[str, int]]]) -> Tuple[int, int]:
    """(64-bit instructions at 4 mod 8, 64-bit instructions) over all functions of a disassembly — without the
    alignment padding the assembler adds, i.e. only meaningful for files without inner .p2align."""
    bad = total = 0
    for seq in sizes.values():
        off = 0
        for _, sz in seq:
            if sz % 8 == 0:
                total += 1
                bad += off % 8 == 4
            off += sz
    return bad, total


def align_file(asm_in: Path, asm_out: Path, scratch_obj: Path, run: int = RUN) -> Dict[str, int]:
    assemble(asm_in, scratch_obj)
    text, stats = align_text(asm_in.read_text(), disassembly_sizes(scratch_obj), run)
    asm_out.write_text(text)
    return stats
