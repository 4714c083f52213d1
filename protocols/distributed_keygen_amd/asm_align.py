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
    choose, per function, where to re-encode an e32 instruction as e64 and where to insert `s_nop 0` so that as few
    64-bit instructions as possible start at 4 mod 8 (a small dynamic programme, see align_text)
    assemble the result (build.py links, bundles and embeds it into the host object)

Where the 32-bit instructions right in front of such a run include a plain VALU operation in its e32 encoding
(v_mov_b32, v_and_b32, v_add_u32, shifts ... on registers and inline constants), that one is re-encoded as e64
instead — the same operation in 8 bytes, which moves everything behind it by 4 bytes at no cost in issue slots; this
is done for runs of any length.  An s_nop costs an issue slot itself, hence the minimum run length for it.  The pass changes no instruction and no
register; it only re-encodes e32 as e64 and adds s_nop 0 between instructions — never inside an inline-asm block, and
never within the four instructions behind an s_getpc_b64: the s_add_u32 / s_addc_u32 that follow it carry sym@rel32 literals whose
addends (+4, +12) assume that they follow it directly.  If the assembly and the disassembly of a function cannot be
matched line by line the function is left as the compiler wrote it.
"""

from __future__ import annotations

import re
import subprocess
from pathlib import Path
from typing import Dict, List, Optional, Tuple

def _llvm_bin() -> Path:
    """The LLVM tools of the ROCm installation (clang as assembler, lld, llvm-objdump, clang-offload-bundler)."""
    import os
    import shutil

    cands = [Path(os.environ.get("ROCM_PATH", "/opt/rocm")) / "lib" / "llvm" / "bin"]
    hipcc = shutil.which("hipcc")
    if hipcc:
        cands.append(Path(hipcc).resolve().parent.parent / "lib" / "llvm" / "bin")
    for c in cands:
        if (c / "llvm-objdump").exists() and (c / "lld").exists():
            return c
    raise RuntimeError("LLVM tools of ROCm not found (looked in " + ", ".join(map(str, cands)) + "); set ROCM_PATH")


LLVM_BIN = _llvm_bin()
NOP_COST = 4.0    # an inserted s_nop is taken when it aligns more than this many 64-bit instructions (3, 5, 8 measured alike)
# Left alone: the instances with 3 limbs per lane (template arguments <K, 3, 29, ...>).  Their blocks are a few dozen
# instructions on ONE dependent chain (the latency geometry); measured, s_nop insertion cost them 3-6 % (14.8 -> 15.3 ms
# and 55.9 -> 59.5 ms per decrypt) and re-encoding alone changed nothing (14.87 -> 14.99, 55.9 -> 55.4).
# ... and the time-sliced instances of the two-wavefront kernel at 9 limbs per lane (template argument PERSISTENT = true):
# they run two and three workgroups per CU, where an inserted s_nop costs an issue slot that a neighbour would have used
# (10 000 ciphertexts, two per CU: 43.8 ms as compiled, 48.6 aligned; one per CU they gain, 46.5 -> 43.7, but that form
# is not chosen anywhere).  The 18-limb time-sliced instances run one workgroup per CU — a wavefront alone on its SIMD —
# and take the pass: 33.5 -> 32.4 ms for 8192 ciphertexts, 42.5 -> 41.1 for 10 000 (profiles/r05_ts_probe_2048*.txt).
SKIP = re.compile(r"ELi3ELi29E|powmod_n2_split_kernelILi\d+ELi9ELi29ELb1E")
# The pass only touches the kernels it was MEASURED to help: the two forms of the N^2 pair kernel at 9 and 18 limbs per
# lane (lone launches -3 ... -14 %) and the generic modexp at 9 and 18 (lone launches -7 ... -9 %: 2.57 -> 2.38 ms for 256
# candidates at key_length 1024, 13.9 -> 12.7 ms for one wide launch at 2048, profiles/r04_sweep_generic*.txt; +0.2 % at
# saturation); every other kernel of the library — combine, verdict, sieve, Jacobi, field, inverse, the 3-limb latency
# instances, the time-sliced instances — is left exactly as the compiler wrote it.
ONLY = re.compile(r"powmod_n2_kernel|powmod_n2_split_kernel|powmod_kernel")
# The pass reads the compiler's assembly text and the disassembler's output; it was validated (CPU tests of the rules,
# the whole GPU parity suite on the aligned library) with this toolchain.  With another one it still falls back per
# function when assembly and disassembly cannot be matched, and build.py prints a note.
VALIDATED_TOOLCHAIN = "ROCm 7.2.0 (AMD clang 22.0.0git)"


def toolchain_note() -> Optional[str]:
    """None if the assembler in use is the validated one, else a sentence for the build log."""
    try:
        out = subprocess.run([str(LLVM_BIN / "clang"), "--version"], capture_output=True, text=True, timeout=30).stdout
    except Exception as exc:  # pragma: no cover
        return f"asm_align: could not query the toolchain ({exc}); validated with {VALIDATED_TOOLCHAIN}"
    m = re.search(r"clang version (\S+)", out)
    ver = m.group(1) if m else "?"
    return None if ver.startswith("22.0.0") else f"asm_align: validated with {VALIDATED_TOOLCHAIN}, running with clang {ver}"
# e32 -> e64 re-encoding: VOP1 / VOP2 / VOPC operations whose VOP3 form takes the same operand text.  Operands:
# VGPRs, inline constants, and at most ONE scalar source (an SGPR or vcc: a VOP3 encoding on this target reads the
# constant bus once and takes no literal; a 4-byte e32 instruction has no literal to begin with).
_promotable = re.compile(
    r"^(v_(?:mov_b32|not_b32|and_b32|or_b32|xor_b32|add_u32|sub_u32|subrev_u32|add_co_u32|sub_co_u32|subrev_co_u32|"
    r"addc_co_u32|subb_co_u32|subbrev_co_u32|lshlrev_b32|lshrrev_b32|ashrrev_i32|max_u32|min_u32|max_i32|min_i32|"
    r"mul_u32_u24|cndmask_b32|cmp_(?:eq|ne|lt|le|gt|ge)_(?:u32|i32)))_e32\s+(.*?)\s*(;.*)?$")
_vgpr_or_const = re.compile(r"^(v\d+|-?\d+)$")
_scalar = re.compile(r"^(s\d+|vcc)$")


def _promoted(line: str) -> Optional[str]:
    """The e64 spelling of an e32 VALU line, or None if it is not one of the plain cases."""
    m = _promotable.match(line.strip())
    if not m:
        return None
    ops = [o.strip() for o in m.group(2).split(",")]
    scalars = set()
    for o in ops:
        if _scalar.match(o):
            scalars.add(o)
        elif not _vgpr_or_const.match(o) or (o.lstrip("-").isdigit() and not -16 <= int(o) <= 64):
            return None
    if len(scalars) > 1:
        return None
    return "\t" + m.group(1) + "_e64 " + ", ".join(ops)


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


_depth = re.compile(r"Depth=(\d+)")


def align_text(asm: str, sizes: Dict[str, List[Tuple[str, int]]], nop_cost: float = NOP_COST, skip=SKIP,
               promote: bool = True, only=None) -> Tuple[str, Dict[str, int]]:
    """Returns the assembly with re-encoded e32 instructions and inserted s_nop 0, and {function: changes made}
    (-1: the function could not be matched with its disassembly and was left alone, -2: its name matches `skip` or
    does not match `only`).

    Per function a two-state dynamic programme over the instruction sequence (state = address mod 8 in {0, 4}):
    a 64-bit instruction at 4 mod 8 costs 1, an inserted s_nop `nop_cost`, both weighted 8^loop depth (the compiler
    annotates blocks with their depth); a 32-bit instruction flips the state, a re-encodable one may keep it
    instead (and then counts as a 64-bit instruction itself)."""
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
        if (skip is not None and skip.search(name)) or (only is not None and not only.search(name)):
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
        size_at: Dict[int, int] = {}
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
            size_at[bi] = dis[k][1]
            k += 1
        if not ok:
            stats[name] = -1
            out.extend(lines[i:j])
            i = j
            continue
        # ---- the items of the dynamic programme
        items: List[Tuple[int, str, float, bool]] = []      # (body index, kind, weight, nop allowed in front)
        in_asm_block = False
        after_getpc = 0          # s_getpc_b64 + s_add_u32/s_addc_u32 with sym@rel32 literals: position dependent
        depth = 0
        for bi, line in enumerate(body):
            s = line.strip()
            if s.startswith(";;#ASMSTART"):
                in_asm_block = True
            elif s.startswith(";;#ASMEND"):
                in_asm_block = False
            if _local_label.match(line) or s.startswith("; %bb."):
                d = _depth.search(line)
                depth = int(d.group(1)) if d else 0
            if s.startswith(".p2align"):
                if int(re.split(r"[\s,]+", s)[1]) >= 3:
                    items.append((bi, "A", 0.0, False))
                continue
            if bi not in size_at:
                continue
            sz = size_at[bi]
            w = float(8 ** min(depth, 3))
            free = not in_asm_block and after_getpc == 0
            if sz % 8 == 0:
                kind = "8"
            elif promote and free and sz == 4 and _promoted(line) is not None:
                kind = "P"
            else:
                kind = "4"
            items.append((bi, kind, w, free))
            after_getpc = 4 if s.startswith("s_getpc_b64") else max(0, after_getpc - 1)
        # ---- forward pass: best[state] = (cost, back pointer)
        INF = float("inf")
        cost = [0.0, INF]                                   # the function starts 256-byte aligned
        back: List[List[Optional[Tuple[int, int, int]]]] = []      # per item, per resulting state: (previous state, nop, promoted)
        for bi, kind, w, nop_ok in items:
            nxt = [INF, INF]
            bp: List[Optional[Tuple[int, int, int]]] = [None, None]
            for p in (0, 1):
                if cost[p] == INF:
                    continue
                for nop in ((0, 1) if nop_ok else (0,)):
                    c0 = cost[p] + (nop_cost * w if nop else 0.0)
                    p1 = p ^ nop
                    options = []
                    if kind == "A":
                        options.append((0, c0, 0))
                    elif kind == "8":
                        options.append((p1, c0 + (w if p1 else 0.0), 0))
                    elif kind == "4":
                        options.append((p1 ^ 1, c0, 0))
                    else:
                        options.append((p1 ^ 1, c0, 0))
                        options.append((p1, c0 + (w if p1 else 0.0), 1))
                    for p2, c2, pr in options:
                        if c2 < nxt[p2]:
                            nxt[p2] = c2
                            bp[p2] = (p, nop, pr)
            cost = nxt
            back.append(bp)
        # ---- backward pass: the decisions
        state = 0 if cost[0] <= cost[1] else 1
        nop_before: Dict[int, bool] = {}
        promoted: Dict[int, bool] = {}
        for idx in range(len(items) - 1, -1, -1):
            prev = back[idx][state]
            assert prev is not None
            p, nop, pr = prev
            if nop:
                nop_before[items[idx][0]] = True
            if pr:
                promoted[items[idx][0]] = True
            state = p
        new_body: List[str] = []
        for bi, line in enumerate(body):
            if nop_before.get(bi):
                new_body.append("\ts_nop 0")
            new_body.append(_promoted(line) if promoted.get(bi) else line)
        stats[name] = len(nop_before) + len(promoted)
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


def align_file(asm_in: Path, asm_out: Path, scratch_obj: Path, nop_cost: float = NOP_COST) -> Dict[str, int]:
    assemble(asm_in, scratch_obj)
    text, stats = align_text(asm_in.read_text(), disassembly_sizes(scratch_obj), nop_cost, only=ONLY)
    asm_out.write_text(text)
    return stats
