"""The build's assembly alignment pass (protocols/distributed_keygen_amd/asm_align.py): what it changes and, more
importantly, what it must not.  Pure text logic — the GPU suite runs on the library built through it."""

from __future__ import annotations

from protocols.distributed_keygen_amd import asm_align as A

SZ = {"s_waitcnt": 4, "v_mad_u64_u32": 8, "v_add_u32_e32": 4, "v_add_u32_e64": 8, "s_getpc_b64": 4, "s_add_u32": 8, "s_addc_u32": 8,
      "s_sleep": 4, "v_fma_f64": 8, "s_add_i32": 4, "s_nop": 4, "v_mov_b32_e32": 4, "v_mov_b32_e64": 8}


def _fn(name, insns):
    """(assembly text, sizes) of a function made of (mnemonic, size[, raw line]) tuples and raw directive lines."""
    lines = [f"{name}:"]
    sizes = []
    for ins in insns:
        if isinstance(ins, str):
            lines.append(ins)
            continue
        lines.append("\t" + (ins[2] if len(ins) > 2 else f"{ins[0]} v0, v1, v2"))
        sizes.append((ins[0], ins[1]))
    lines.append(".Lfunc_end0:")
    return "\n".join(lines), {name: sizes}


def _layout(text):
    """[(mnemonic, offset)] of the instruction lines of a processed function."""
    off, out = 0, []
    for line in text.split("\n")[1:]:
        s = line.strip()
        if s.startswith(".p2align"):
            a = 1 << int(s.split()[1].rstrip(","))
            off = (off + a - 1) // a * a
        if not A._is_insn(line):
            continue
        m = line.split()[0]
        out.append((m, off))
        off += SZ[m]
    return out


def _misaligned(text):
    return sum(1 for m, off in _layout(text) if SZ[m] == 8 and off % 8 == 4)


MAD = ("v_mad_u64_u32", 8)
SALU = ("s_add_i32", 4, "s_add_i32 s0, s1, s2")


def test_a_long_misaligned_run_gets_an_s_nop_and_a_short_one_does_not():
    asm, sizes = _fn("k", [("s_waitcnt", 4, "s_waitcnt vmcnt(0)"), *[MAD] * 6, SALU, SALU, SALU, *[MAD] * 2, SALU])
    out, stats = A.align_text(asm, sizes, nop_cost=4.0, skip=None)
    assert out.count("s_nop 0") == 1 and stats["k"] == 1
    lay = _layout(out)
    assert all(off % 8 == 0 for m, off in lay[:8] if m == "v_mad_u64_u32")          # the six behind the s_waitcnt
    assert _misaligned(out) == 2                                                     # two at the end: not worth a slot
    # with a free s_nop everything is aligned; with an expensive one nothing is touched
    assert _misaligned(A.align_text(asm, sizes, nop_cost=0.5, skip=None)[0]) == 0
    assert A.align_text(asm, sizes, nop_cost=100.0, skip=None)[0] == asm


def test_an_e32_instruction_in_front_is_re_encoded_instead_of_an_s_nop():
    """v_add_u32_e32 -> v_add_u32_e64: the same operation in 8 bytes moves everything behind it by 4 at no issue slot."""
    asm, sizes = _fn("k", [("v_add_u32_e32", 4, "v_add_u32_e32 v3, v4, v5"), *[MAD] * 2])
    out, stats = A.align_text(asm, sizes, skip=None)
    assert "v_add_u32_e64 v3, v4, v5" in out and "s_nop" not in out and _misaligned(out) == 0 and stats["k"] == 1
    # the instances with 3 limbs per lane are skipped by name
    asm3, sizes3 = _fn("_ZN2mx6kernelILi32ELi3ELi29EEEv", [("v_mov_b32_e32", 4, "v_mov_b32_e32 v1, v2"), *[MAD] * 9])
    out3, stats3 = A.align_text(asm3, sizes3)
    assert out3 == asm3 and stats3["_ZN2mx6kernelILi32ELi3ELi29EEEv"] == -2
    # ... and so are the time-sliced instances of the two-wavefront pair kernel; the build's allow-list (ONLY) leaves
    # every kernel alone that the pass was not measured to help
    body = [("v_mov_b32_e32", 4, "v_mov_b32_e32 v1, v2"), *[MAD] * 9]
    ts = "_ZN2mx22powmod_n2_split_kernelILi8ELi9ELi29ELb1ELb1EEEvNS_12PowmodN2ArgsE"
    plain = "_ZN2mx22powmod_n2_split_kernelILi8ELi9ELi29ELb0ELb1EEEvNS_12PowmodN2ArgsE"
    generic = "_ZN2mx13powmod_kernelILi8ELi9ELi29ELb0ELb0EEEvNS_10PowmodArgsE"
    generic_latency = "_ZN2mx13powmod_kernelILi32ELi3ELi29ELb0ELb1EEEvNS_10PowmodArgsE"
    combine = "_ZN2mx14combine_kernelILi16ELi9ELi29EEEvNS_11CombineArgsE"
    one_wave = "_ZN2mx16powmod_n2_kernelILi4ELi18ELi29ELb1EEEvNS_12PowmodN2ArgsE"
    for name, touched in ((ts, False), (plain, True), (generic, True), (generic_latency, False), (combine, False), (one_wave, True)):
        a, sz = _fn(name, body)
        o, st = A.align_text(a, sz, only=A.ONLY)
        assert (o != a) == touched and (st[name] > 0) == touched and (touched or st[name] == -2), name
    assert A.toolchain_note() is None or "validated with" in A.toolchain_note()
    # the optimum over a sequence: re-encode the first, keep the second (which would undo it)
    asm2, sizes2 = _fn("k", [("v_mov_b32_e32", 4, "v_mov_b32_e32 v1, v2"), *[MAD] * 4, ("v_mov_b32_e32", 4, "v_mov_b32_e32 v3, v4"), ("v_mov_b32_e32", 4, "v_mov_b32_e32 v5, v6"), *[MAD] * 4])
    out2, _ = A.align_text(asm2, sizes2, skip=None)
    assert _misaligned(out2) == 0 and "s_nop" not in out2


def test_only_plain_operand_forms_are_re_encoded():
    ok = {"v_mov_b32_e32 v1, v2": "v_mov_b32_e64 v1, v2", "v_and_b32_e32 v1, s4, v2": "v_and_b32_e64 v1, s4, v2",
          "v_cndmask_b32_e32 v0, v1, v2, vcc": "v_cndmask_b32_e64 v0, v1, v2, vcc", "v_cmp_eq_u32_e32 vcc, v0, v1": "v_cmp_eq_u32_e64 vcc, v0, v1",
          "v_addc_co_u32_e32 v0, vcc, v1, v2, vcc": "v_addc_co_u32_e64 v0, vcc, v1, v2, vcc", "v_lshlrev_b32_e32 v0, 2, v1  ; x": "v_lshlrev_b32_e64 v0, 2, v1"}
    for src, want in ok.items():
        assert A._promoted("\t" + src) == "\t" + want
    for src in ("v_cndmask_b32_e32 v0, s1, v2, vcc",            # two reads of the constant bus
                "v_add_u32_e32 v0, 100, v1",                    # a literal (not an inline constant)
                "v_and_b32_e32 v17, 0x1fffffff, v17", "v_readfirstlane_b32 s0, v1", "v_mov_b32_dpp v0, v1 row_shr:1",
                "v_mad_u64_u32 v[0:1], s[0:1], v2, v3, v[4:5]", "s_add_i32 s0, s1, s2", "v_add_u32_e32 v0, exec_lo, v1"):
        assert A._promoted("\t" + src) is None, src


def test_nothing_changes_right_behind_s_getpc():
    """s_add_u32 / s_addc_u32 behind s_getpc_b64 carry sym@rel32@lo+4 / @hi+12: anything that moves them away from the
    address s_getpc returned makes every load from that table 4 bytes off (the first version of the pass did)."""
    asm, sizes = _fn("k", [("s_getpc_b64", 4, "s_getpc_b64 s[0:1]"), ("s_add_u32", 8, "s_add_u32 s0, s0, tab@rel32@lo+4"),
                            ("s_addc_u32", 8, "s_addc_u32 s1, s1, tab@rel32@hi+12"), *[MAD] * 12])
    out, stats = A.align_text(asm, sizes, nop_cost=1.0, skip=None)
    lines = [l.strip() for l in out.split("\n")]
    g = lines.index("s_getpc_b64 s[0:1]")
    assert lines[g + 1].startswith("s_add_u32") and lines[g + 2].startswith("s_addc_u32")
    assert stats["k"] == 1 and not g < lines.index("s_nop 0") <= g + 4          # in front of the sequence or behind it, never inside


def test_inline_asm_blocks_and_unmatched_functions_are_left_alone():
    body = [("s_waitcnt", 4, "s_waitcnt lgkmcnt(0)"), "\t;;#ASMSTART", *[("v_fma_f64", 8)] * 6, ("v_mov_b32_e32", 4, "v_mov_b32_e32 v1, v2"), "\t;;#ASMEND",
            SALU, *[MAD] * 6]
    asm, sizes = _fn("k", body)
    out, _ = A.align_text(asm, sizes, nop_cost=1.0, skip=None)
    lines = [l.strip() for l in out.split("\n")]
    a, b = lines.index(";;#ASMSTART"), lines.index(";;#ASMEND")
    assert lines[a:b + 1] == [l.strip() for l in asm.split("\n")][a:b + 1] or lines[a + 1:b] == [l.strip() for l in asm.split("\n") if "v_fma_f64" in l or "v_mov_b32_e32" in l]
    assert "s_nop 0" not in lines[a:b] and "v_mov_b32_e64 v1, v2" not in lines
    # a function whose lines do not match the disassembly is not touched
    bad = {"k": [("s_waitcnt", 4), ("v_something_else", 8)]}
    out2, stats2 = A.align_text(asm, bad)
    assert out2 == asm and stats2["k"] == -1
    # functions without sizes (other sections, data labels) pass through
    assert A.align_text("data_label:\n\t.long 5\n", {})[0] == "data_label:\n\t.long 5\n"


def test_p2align_resets_the_state_and_padding_in_the_disassembly_is_tolerated():
    asm, _ = _fn("k", [("s_waitcnt", 4, "s_waitcnt vmcnt(0)"), "\t.p2align\t6", *[MAD] * 6])
    sizes = {"k": [("s_waitcnt", 4), *[("s_nop", 4)] * 15, *[MAD] * 6]}             # the assembler's padding shows up as s_nop
    out, stats = A.align_text(asm, sizes, skip=None)
    assert stats["k"] == 0 and out == asm


def test_inner_loops_win_over_straight_line_code():
    """Weights are 8^depth: a conflict between a loop body and the code in front of it is settled for the loop."""
    asm, sizes = _fn("k", [*[MAD] * 5, SALU, ".LBB0_1:                                ; =>This Inner Loop Header: Depth=1", *[MAD] * 3,
                            ("s_waitcnt", 4, "s_waitcnt vmcnt(0)"), SALU])
    out, _ = A.align_text(asm, sizes, nop_cost=6.0, skip=None)
    lay = _layout(out)
    assert all(off % 8 == 0 for m, off in lay[-5:-2])          # the three in the loop
    assert out.count("s_nop 0") == 1


def test_skip_pattern_names_the_three_limb_instances_only():
    assert A.SKIP.search("_ZN2mx22powmod_n2_split_kernelILi32ELi3ELi29ELb0EEEvNS_12PowmodN2ArgsE")
    assert A.SKIP.search("_ZN2mx22powmod_n2_split_kernelILi64ELi3ELi29ELb0EEEvNS_12PowmodN2ArgsE")
    assert not A.SKIP.search("_ZN2mx22powmod_n2_split_kernelILi8ELi9ELi29ELb0EEEvNS_12PowmodN2ArgsE")
    assert not A.SKIP.search("_ZN2mx16powmod_n2_kernelILi4ELi18ELi29EEEvNS_12PowmodN2ArgsE")
    assert not A.SKIP.search("_ZN2mx13jacobi_kernelILi3EEEvNS_10JacobiArgsE")
