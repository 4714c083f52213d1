"""The build's assembly alignment pass (protocols/distributed_keygen_amd/asm_align.py): where it inserts s_nop 0 and,
more importantly, where it must not.  Pure text logic — the GPU suite runs on the library built through it."""

from __future__ import annotations

import re

from protocols.distributed_keygen_amd import asm_align as A


def _fn(name, insns):
    """(assembly text, sizes) of a function made of (mnemonic, size[, raw line]) tuples."""
    lines = [f"{name}:"]
    sizes = []
    for ins in insns:
        if isinstance(ins, str):                   # directive / label / comment line
            lines.append(ins)
            continue
        mnem, size = ins[0], ins[1]
        lines.append("\t" + (ins[2] if len(ins) > 2 else f"{mnem} v0, v1, v2"))
        sizes.append((mnem, size))
    lines.append(".Lfunc_end0:")
    return "\n".join(lines), {name: sizes}


def _layout(text, sizes_by_mnemonic):
    """[(mnemonic, offset)] of the instruction lines of a processed function (s_nop = 4 bytes)."""
    off, out = 0, []
    for line in text.split("\n")[1:]:
        if not A._is_insn(line):
            continue
        m = line.split()[0]
        out.append((m, off))
        off += 4 if m == "s_nop" else sizes_by_mnemonic[m]
    return out


SZ = {"s_waitcnt": 4, "v_mad_u64_u32": 8, "v_add_u32_e32": 4, "s_getpc_b64": 4, "s_add_u32": 8, "s_addc_u32": 8, "v_mov_b32_dpp": 8,
      "s_sleep": 4, "v_fma_f64": 8}


def test_a_misaligned_run_gets_one_nop_and_short_runs_none():
    mad = ("v_mad_u64_u32", 8)
    asm, sizes = _fn("k", [("s_waitcnt", 4, "s_waitcnt vmcnt(0)"), *[mad] * 6, ("v_add_u32_e32", 4), *[mad] * 6,
                            ("v_add_u32_e32", 4), ("v_add_u32_e32", 4), ("v_add_u32_e32", 4), *[mad] * 2, ("v_add_u32_e32", 4)])
    out, stats = A.align_text(asm, sizes, run=5, skip=None)
    lay = _layout(out, SZ)
    # first run: misaligned behind the 4-byte s_waitcnt -> nop; second run: behind waitcnt + nop + 6 mads + one add: aligned at 4+4+48+4 = 60?  no: 60 % 8 == 4 -> nop
    assert stats["k"] == 2
    for m, off in lay:
        if m == "v_mad_u64_u32" and lay.index((m, off)) < 16:
            assert off % 8 == 0, (m, off)
    # the run of two at the end is too short to be worth a nop
    tail = [x for x in lay if x[0] == "v_mad_u64_u32"][-2:]
    assert tail[0][1] % 8 == 4
    assert out.count("s_nop 0") == 2


def test_nothing_is_inserted_right_behind_s_getpc():
    """s_add_u32 / s_addc_u32 behind s_getpc_b64 carry sym@rel32@lo+4 / @hi+12: an instruction in between would move the
    literals away from the address s_getpc returned and every load from that table would be 4 bytes off."""
    mad = ("v_mad_u64_u32", 8)
    asm, sizes = _fn("k", [("s_getpc_b64", 4, "s_getpc_b64 s[0:1]"), ("s_add_u32", 8, "s_add_u32 s0, s0, tab@rel32@lo+4"),
                            ("s_addc_u32", 8, "s_addc_u32 s1, s1, tab@rel32@hi+12"), *[mad] * 8])
    out, stats = A.align_text(asm, sizes, run=3, skip=None)
    lines = [l.strip() for l in out.split("\n")]
    g = lines.index("s_getpc_b64 s[0:1]")
    assert lines[g + 1].startswith("s_add_u32") and lines[g + 2].startswith("s_addc_u32")
    assert stats["k"] == 1 and lines.index("s_nop 0") >= g + 5          # the run is fixed further down instead


def test_inline_asm_blocks_unknown_functions_and_skipped_instances_are_left_alone():
    mad = ("v_mad_u64_u32", 8)
    body = [("s_waitcnt", 4, "s_waitcnt lgkmcnt(0)"), "\t;;#ASMSTART", *[("v_fma_f64", 8)] * 6, "\t;;#ASMEND", ("v_add_u32_e32", 4), *[mad] * 6]
    asm, sizes = _fn("_ZN2mx6kernelILi32ELi3ELi29EEEv", body)
    out, stats = A.align_text(asm, sizes, run=3)
    assert stats["_ZN2mx6kernelILi32ELi3ELi29EEEv"] == -2 and out == asm          # the 3-limb instances are skipped by name
    asm, sizes = _fn("_ZN2mx6kernelILi8ELi9ELi29EEEv", body)
    out, stats = A.align_text(asm, sizes, run=3)
    lines = [l.strip() for l in out.split("\n")]
    a, b = lines.index(";;#ASMSTART"), lines.index(";;#ASMEND")
    assert "s_nop 0" not in lines[a:b] and lines[a - 1].startswith("s_waitcnt")   # nothing in or right in front of the asm block's run
    assert stats["_ZN2mx6kernelILi8ELi9ELi29EEEv"] == 0                           # ... and the mads behind it are aligned again (4 + 48 + 4)
    # a function whose lines do not match the disassembly is not touched
    bad = {"_ZN2mx6kernelILi8ELi9ELi29EEEv": [("s_waitcnt", 4), ("v_something_else", 8)]}
    out2, stats2 = A.align_text(asm, bad, run=3)
    assert out2 == asm and stats2["_ZN2mx6kernelILi8ELi9ELi29EEEv"] == -1
    # functions without sizes (other sections, data labels) pass through
    assert A.align_text("data_label:\n\t.long 5\n", {}, run=3)[0] == "data_label:\n\t.long 5\n"


def test_p2align_resets_the_offset_and_padding_in_the_disassembly_is_tolerated():
    mad = ("v_mad_u64_u32", 8)
    asm, _ = _fn("k", [("s_waitcnt", 4, "s_waitcnt vmcnt(0)"), "\t.p2align\t6", *[mad] * 6])
    sizes = {"k": [("s_waitcnt", 4), *[("s_nop", 4)] * 15, *[mad] * 6]}             # the assembler's padding shows up as s_nop
    out, stats = A.align_text(asm, sizes, run=3, skip=None)
    assert stats["k"] == 0 and out == asm


def test_skip_pattern_names_the_three_limb_instances_only():
    assert A.SKIP.search("_ZN2mx22powmod_n2_split_kernelILi32ELi3ELi29ELb0EEEvNS_12PowmodN2ArgsE")
    assert A.SKIP.search("_ZN2mx22powmod_n2_split_kernelILi64ELi3ELi29ELb0EEEvNS_12PowmodN2ArgsE")
    assert not A.SKIP.search("_ZN2mx22powmod_n2_split_kernelILi8ELi9ELi29ELb0EEEvNS_12PowmodN2ArgsE")
    assert not A.SKIP.search("_ZN2mx16powmod_n2_kernelILi4ELi18ELi29EEEvNS_12PowmodN2ArgsE")
    assert not A.SKIP.search("_ZN2mx13jacobi_kernelILi3EEEvNS_10JacobiArgsE")
    assert re.compile(A.SKIP.pattern)
