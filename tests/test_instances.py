"""Every kernel instance the modexp launchers can select has a parity case (CPU: geometry queries only).

VERDICT r01 "What's weak" 1: the wide pair kernel at K = 8 / 16 was launched by the product
(key_length 3072 / 4096 at batch >= 3840) without ever having been parity-tested.  This test makes
that class of gap impossible: it enumerates the instances through the library's own geometry
queries and checks that tests/instance_cases.py (run on the GPU by test_gpu_instances.py) pins
each of them.
"""

from __future__ import annotations

from pathlib import Path

import pytest

import instance_cases as ic


def test_every_reachable_instance_has_a_parity_case():
    from protocols.distributed_keygen_amd import _lib

    lib = _lib.lib()
    reachable = ic.reachable_instances(lib)
    covered = {ic.case_instance(lib, c) for c in ic.ALL_CASES}
    assert None not in covered, "a case asks for a geometry the library refuses"
    missing = reachable - covered
    assert not missing, f"kernel instances without a parity case: {sorted(missing)}"
    # the enumeration itself must see the instances VERDICT r01 named, and the full template grid
    assert ("n2", 8, 18, 1, 0, 0) in reachable and ("n2", 16, 18, 1, 0, 0) in reachable
    n2 = [r for r in reachable if r[0] == "n2"]
    for wpg in (1, 2):
        assert {k for _, k, l, w, fr, ts in n2 if l == 9 and w == wpg} == {1, 2, 4, 8, 16, 32}
        assert {k for _, k, l, w, fr, ts in n2 if l == 18 and w == wpg} == {1, 2, 4, 8, 16}
    assert {k for _, k, l, w, fr, ts in n2 if l == 3 and w == 2} == {1, 2, 4, 8, 16, 32, 64}
    assert not {k for _, k, l, w, fr, ts in n2 if l == 3 and w == 1}     # latency geometries are split only
    assert {k for _, k, l, w, fr, ts in n2 if w == 4} == {16, 32, 64} and all(l == 3 and fr and not ts for _, k, l, w, fr, ts in n2 if w == 4)   # four-wavefront form (round 6)
    # the friendly-modulus and the time-sliced instances are instances of their own (VERDICT r03 "weak" 1b)
    assert {(k, l, w) for _, k, l, w, fr, ts in n2 if fr and l != 3} == {(8, 9, 2), (16, 9, 2), (4, 18, 1), (8, 18, 1)}
    assert {(k, l, w, fr) for _, k, l, w, fr, ts in n2 if ts} >= {(8, 9, 2, 1), (8, 9, 2, 0), (16, 9, 2, 1)}
    assert all(w == 2 and ((l == 9 and k <= 16) or (l == 18 and k in (4, 8))) for _, k, l, w, fr, ts in n2 if ts)
    assert {(k, l) for _, k, l, w, fr, ts in n2 if ts and l == 18} == {(4, 18), (8, 18)}          # round 5
    assert all(fr for _, k, l, w, fr, ts in n2 if l == 3)
    for kind in ("generic-sliding", "generic-fixed"):
        assert {k for kd, k, l, *w in reachable if kd == kind and l == 9} == {1, 2, 4, 8, 16, 32, 64}
        assert {k for kd, k, l, *w in reachable if kd == kind and l == 18} == {1, 2, 4, 8, 16, 32}
        assert {k for kd, k, l, *w in reachable if kd == kind and l == 3} == {1, 2, 4, 8, 16, 32, 64}       # latency instances
    assert {k for kd, k, l, *w in reachable if kd == "generic-bi"} == {4, 8, 16, 32, 64}                   # bipartite form (round 5)


def test_auto_launch_shapes_match_the_measured_crossovers():
    """The library's choice for ONE launch on an idle GPU follows the crossovers tools/sweep_shapes.py and tools/ts_probe.py
    measured (profiles/r03_sweep_shapes.txt, r05_ts_probe_*.txt): latency geometry on two wavefronts for a handful of
    ciphertexts, then 9 and 18 limbs per lane still on two wavefronts while that buys occupancy — time-sliced just above
    one workgroup per CU —, the one-wavefront wide kernel once a launch fills the machine on its own."""
    from protocols.distributed_keygen_amd import _lib

    lib = _lib.lib()
    shape = lambda bits, batch: ic.case_instance(lib, ("n2", bits, 0, batch, 0, 0))[:4]
    for batch, want in ((1, (32, 3, 4)), (64, (32, 3, 4)), (512, (32, 3, 4)), (513, (32, 3, 2)), (1000, (32, 3, 2)), (2000, (32, 3, 2)), (3000, (8, 9, 2)), (4096, (8, 9, 2)),
                        (6144, (4, 18, 2)), (8192, (4, 18, 2)), (10000, (4, 18, 2)), (12288, (4, 18, 2)), (16384, (4, 18, 1)),
                        (20000, (4, 18, 2)), (24576, (4, 18, 2)), (30000, (4, 18, 1))):
        assert shape(2051, batch) == ("n2",) + want, batch
    for batch, want in ((1, (64, 3, 4)), (256, (64, 3, 4)), (257, (64, 3, 2)), (512, (64, 3, 2)), (1024, (64, 3, 2)), (2048, (16, 9, 2)), (4096, (8, 18, 2)), (16000, (8, 18, 1))):
        assert shape(4099, batch) == ("n2",) + want, batch
    assert shape(1027, 256) == ("n2", 16, 3, 4) and shape(1027, 1025) == ("n2", 16, 3, 2) and shape(1027, 1000000)[3] == 1
    # an explicit argument pins that half of the choice
    assert ic.case_instance(lib, ("n2", 2051, 18, 10, 0, 0)) == ("n2", 4, 18, 2, 0, 0)
    assert ic.case_instance(lib, ("n2", 2051, 0, 10, 0, 1)) == ("n2", 8, 9, 1, 0, 0)
    assert ic.case_instance(lib, ("n2", 2051, 18, 40000, 0, 1)) == ("n2", 4, 18, 1, 1, 0)        # friendly one-wavefront instance
    assert ic.case_instance(lib, ("n2", 2075, 18, 40000, 0, 1)) == ("n2", 4, 18, 1, 0, 0)        # no room: the plain one
    assert ic.case_instance(lib, ("n2", 2051, 0, 10000, 0, 0)) == ("n2", 4, 18, 2, 0, 1)         # time-sliced, wide (round 5)
    assert ic.case_instance(lib, ("n2", 2051, 9, 10000, 0, 0)) == ("n2", 8, 9, 2, 1, 1)          # time-sliced, friendly
    assert ic.case_instance(lib, ("n2", 2051, 3, 10, 0, 1)) is None          # the latency geometry has no one-wavefront form


def test_time_sliced_launches_where_they_were_measured_to_pay():
    """The time-sliced form of the two-wavefront kernel (resident workgroups taking segment units from a queue) is
    chosen just above a capacity step of a lone launch and nowhere else (tools/ts_probe.py,
    profiles/r05_ts_probe_2048.txt, _4096.txt): at 18 limbs per lane, one workgroup per CU, in 8 or 12 units per group
    (2 where three half-launches are the best there is); an explicit one-wavefront shape never is."""
    import ctypes

    from protocols.distributed_keygen_amd import _lib

    lib = _lib.lib()

    def sliced(bits, batch, lpl=0, wpg=0):
        r, u = ctypes.c_int(), ctypes.c_int()
        assert lib.mx_nsquare_launch_timesliced(bits, batch, lpl, wpg, r, u) == 0
        return r.value, u.value

    for batch in (1, 1000, 4096, 8192, 14336, 16384, 24576, 32768, 40000):
        assert sliced(2051, batch) == (0, 0), batch
    for batch in (9216, 10000, 10240, 11264, 13312, 17408, 18432):      # 37.6 / 40.9 / 41.1 / 45.2 / 53.3 / 69.9 / 74.0 ms
        assert sliced(2051, batch) == (1, 8), batch
    assert sliced(2051, 8704) == (1, 12) and sliced(2051, 12288) == (1, 2) and sliced(2051, 20000) == (1, 2)
    assert sliced(2051, 4608) == (1, 8)
    assert sliced(4099, 5120) == (1, 8) and sliced(4099, 4352) == (1, 12) and sliced(4099, 2304) == (1, 8) and sliced(4099, 2048) == (0, 0)
    assert sliced(2051, 10000, 18, 0) == (1, 8) and sliced(2051, 10000, 0, 1) == (0, 0) and sliced(2051, 10000, 9, 2) == (2, 2)
    assert sliced(8195, 3000) == (0, 0)                   # groups of 32 lanes have no time-sliced instance
    assert lib.mx_nsquare_launch_timesliced(2051, 0, 0, 0, ctypes.c_int(), ctypes.c_int()) == -1


def test_shape_for_launches_that_share_the_machine_as_pieces():
    """mx_nsquare_pieces_shape: the plain form for launches that are in flight together (Engine.saturating_shape below the
    saturating sizes) — never the time-sliced choice of a lone launch; measured with four pieces in flight at key_length
    2048: 4 x 1250 and 4 x 5000 best at 18 limbs on two wavefronts (152 k/s, 280 k/s), 4 x 2500 at 9 limbs (215 against 179)."""
    import ctypes

    from protocols.distributed_keygen_amd import _lib

    lib = _lib.lib()

    def pieces(bits, total, lpl=0, wpg=0):
        l, w = ctypes.c_int(), ctypes.c_int()
        assert lib.mx_nsquare_pieces_shape(bits, total, lpl, wpg, l, w) == 0
        return l.value, w.value

    assert pieces(2051, 5000) == (18, 2) and pieces(2051, 10000) == (9, 2) and pieces(2051, 12288) == (9, 2)
    assert pieces(2051, 20000) == (18, 2) and pieces(2051, 30000) == (18, 1)
    assert pieces(4099, 8192) == (18, 1) and pieces(4099, 8192, 0, 2) == (18, 2) and pieces(4099, 2048) == (9, 2)
    assert pieces(2051, 10000, 18, 0)[0] == 18 and pieces(1027, 2048) == (3, 2)
    assert lib.mx_nsquare_pieces_shape(2051, 0, 0, 0, ctypes.c_int(), ctypes.c_int()) == -1
    assert lib.mx_nsquare_pieces_shape(2051, 100, 7, 0, ctypes.c_int(), ctypes.c_int()) == -1


def test_no_kernel_has_a_private_segment():
    """Every shipped instance of the three modexp kernels runs without scratch memory: private_segment_fixed_size 0 and no
    spilled vector register in the kernel descriptors of the BUILT library (tools/scratch_report.py reads the code
    objects embedded in libmxpaillier.so).  Round 3 shipped the 18-limb pair kernel with 28 spilled registers and the
    time-sliced instances with 59-75 (VERDICT r03 "weak" 2), and the 257-word Jacobi instances with 204 B / 1848 B (their
    top limbs live in LDS now)."""
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root / "tools"))
    import scratch_report

    from protocols.distributed_keygen_amd import _lib

    rows = scratch_report.kernels_of_library(_lib.LIB_PATH)
    names = scratch_report.demangle([r[0] for r in rows])
    assert len(rows) > 100, "kernel metadata of the library not found"
    modexp = [r for r in rows if "powmod" in names[r[0]]]
    assert len(modexp) >= 60, len(modexp)
    for tmpl in ("mx::powmod_n2_kernel<4, 18, 29, true>", "mx::powmod_n2_kernel<4, 18, 29, false>", "mx::powmod_n2_split_kernel<8, 9, 29, true, true>", "mx::powmod_kernel<8, 9, 29, false, false>", "mx::powmod_kernel<32, 3, 29, false, true>",
                 "mx::powmod_n2_split_kernel<32, 3, 29, false, true>"):
        assert any(tmpl in names[r[0]] for r in modexp), tmpl
    offenders = [(names[r[0]], r[1], r[2]) for r in rows if r[1] or r[2]]
    assert not offenders, offenders          # nor any other kernel of the library (round 4: the 257-word Jacobi instances too)


def test_library_staleness_is_decided_by_content_not_by_modification_time():
    """build.needs_build(): a checkout or a copy of the tree renews modification times without changing a byte — the
    library is stale only if a source differs from the digest the build stamped (protocols/.../build/sources.sha256)."""
    import os

    from protocols.distributed_keygen_amd import build as B

    if not B.STAMP.exists():
        pytest.skip("library built before the stamp existed")
    assert B.STAMP.read_text().strip() == B.sources_digest(), "libmxpaillier.so was not built from the sources in the tree"
    assert not B.needs_build()
    header = B.HEADERS[0]
    st = header.stat()
    try:
        os.utime(header, None)                      # newer than the library, same bytes
        assert not B.needs_build()
    finally:
        os.utime(header, (st.st_atime, st.st_mtime))


@pytest.fixture(scope="module")
def library_disassembly(tmp_path_factory):
    """{kernel symbol: [instruction text, ...]} of every code object embedded in the library under test (MX_LIBRARY or the
    in-tree build), from llvm-objdump."""
    import os
    import subprocess
    import sys

    sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))
    import scratch_report

    from protocols.distributed_keygen_amd import asm_align, build as B

    lib = Path(os.environ.get("MX_LIBRARY") or B.LIB)
    objdump = asm_align.LLVM_BIN / "llvm-objdump"
    if not objdump.exists() or not lib.exists():
        pytest.skip("no llvm-objdump / library")
    tmp = tmp_path_factory.mktemp("disasm")
    fns = {}
    for i, elf in enumerate(scratch_report.code_objects_of_library(lib)):
        f = tmp / f"co{i}.elf"
        f.write_bytes(elf)
        text = subprocess.run([str(objdump), "-d", str(f)], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in text.splitlines():
            if line.endswith(">:"):
                cur = fns.setdefault(line.split("<", 1)[1][:-2], [])
            elif cur is not None and line.startswith("\t"):
                cur.append(line.split("//", 1)[0].strip())
        f.unlink()
    return fns


def test_short_kernels_raise_their_wave_priority_and_modexp_kernels_do_not(library_disassembly):
    """csrc/mx_prio.hpp: the Jacobi filter, selection, R mod N, verdict and recombination kernels start with
    `s_setprio 3` (they share SIMDs with other steps' modexp wavefronts); no modexp kernel touches its
    priority.  Read from the disassembly of the library as built."""
    short = ("jacobi_kernel", "jacobi_fallback_kernel", "select_first_kernel", "rmodn_kernel", "verdict_kernel", "combine_kernel")
    raised, modexp_with_prio, seen_short = set(), set(), set()
    for fn, insns in library_disassembly.items():
        is_short = any(s in fn for s in short)
        if is_short:
            seen_short.add(fn)
        for ins in insns:
            if ins.startswith("s_setprio"):
                if is_short:
                    assert ins == "s_setprio 3", (fn, ins)
                    raised.add(fn)
                elif "powmod" in fn:
                    modexp_with_prio.add(fn)
    assert len(seen_short) >= 20, sorted(seen_short)[:5]
    assert seen_short == raised, sorted(seen_short - raised)[:5]
    assert not modexp_with_prio, sorted(modexp_with_prio)[:5]


def test_time_sliced_instances_wait_for_the_l2_write_back_before_they_publish(library_disassembly):
    """The hand-over of a group between wavefront pairs that may sit on different XCDs (csrc/mx_powmod_n2_split.hpp:
    ts_release_agent / ts_acquire_agent).  Rounds 3-4 shipped the compiler's sequence for a release store behind an atomic
    whose result had been waited for — `buffer_wbl2 sc1 ; s_waitcnt lgkmcnt(0) ; global_store` — which does not wait for
    the write-back, and 1 group in ~1000 hot hand-overs was read half-written.  In EVERY time-sliced instance of the
    built library: each `buffer_wbl2 sc1` is directly followed by an s_waitcnt that drains vmcnt, both wavefronts of a
    pair have one (A before the token that lets B push, B before it reserves the ring entry), the taker invalidates
    (`buffer_inv sc1`) and waits; the plain instances have neither.  tools/prove_handover_guard.sh shows this test failing on a
    -DMX_DEV_TS_COMPILER_RELEASE build; the behavioural test is tests/test_gpu_handover.py."""
    import re

    ts = {fn: ins for fn, ins in library_disassembly.items() if re.search(r"powmod_n2_split_kernelILi\d+ELi\d+ELi\d+ELb1E", fn)}
    plain = {fn: ins for fn, ins in library_disassembly.items() if re.search(r"powmod_n2_split_kernelILi\d+ELi\d+ELi\d+ELb0E", fn)}
    assert len(ts) >= 9 and len(plain) >= 18, (len(ts), len(plain))
    assert {tuple(map(int, re.search(r"ILi(\d+)ELi(\d+)E", fn).groups())) for fn in ts} >= {(8, 9), (16, 9), (4, 18), (8, 18)}
    store_like = re.compile(r"^(global|flat|buffer|scratch)_(store|atomic)")
    for fn, insns in ts.items():
        wb = [i for i, ins in enumerate(insns) if ins.startswith("buffer_wbl2")]
        assert len(wb) == 2, (fn, len(wb))
        for i in wb:
            assert insns[i] == "buffer_wbl2 sc1", (fn, insns[i])
            nxt = insns[i + 1]
            assert nxt.startswith("s_waitcnt") and "vmcnt(0)" in nxt, (fn, insns[i - 2:i + 4])
            # ... and this wavefront's own stores were complete before the write-back was asked for
            prev = insns[i - 1]
            assert prev.startswith("s_waitcnt") and "vmcnt(0)" in prev, (fn, insns[i - 2:i + 2])
        # no store or atomic to device memory may sit between a write-back and its wait (vacuous with the assertion
        # above; kept for a compiler that schedules something in between)
        for i in wb:
            j = i + 1
            while not insns[j].startswith("s_waitcnt"):
                assert not store_like.match(insns[j]), (fn, insns[i:j + 1])
                j += 1
        inv = [i for i, ins in enumerate(insns) if ins.startswith("buffer_inv")]
        assert inv, fn
        for i in inv:
            assert insns[i] == "buffer_inv sc1" and insns[i + 1].startswith("s_waitcnt") and "vmcnt(0)" in insns[i + 1], (fn, insns[i:i + 2])
    for fn, insns in plain.items():
        assert not any(ins.startswith(("buffer_wbl2", "buffer_inv")) for ins in insns), fn


def test_developer_build_switches_live_in_one_header_and_the_shipped_build_sets_none():
    """csrc/mx_dev.hpp: every macro a kernel source branches on is an MX_DEV_ switch documented there; build.py's flags
    define none of them (VERDICT r05 "weak" 11)."""
    import re

    from protocols.distributed_keygen_amd import build as B

    assert not [f for f in B.FLAGS if f.startswith("-DMX")]
    documented = set(re.findall(r"^//\s+(MX_DEV_[A-Z_0-9]+)", (B.CSRC / "mx_dev.hpp").read_text(), flags=re.M))
    assert len(documented) >= 8
    used = set()
    for src in list(B.CSRC.glob("*.hpp")) + list(B.CSRC.glob("*.hip")):
        text = src.read_text()
        for m in re.finditer(r"^\s*#\s*(?:if|ifdef|ifndef|elif)\b(.*)$", text, flags=re.M):
            names = set(re.findall(r"\b[A-Z][A-Z_0-9]{3,}\b", m.group(1)))
            assert all(nm.startswith("MX_DEV_") for nm in names), (src.name, m.group(0))
            used |= names
        used |= set(re.findall(r"\bMX_DEV_[A-Z_0-9]+\b", text)) if src.name != "mx_dev.hpp" else set()
    assert used - {"MX_DEV_BUILD"} <= documented, used - documented
