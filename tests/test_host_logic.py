"""CPU tests of the host layer: C-ABI exports, limb packing, and the mirrors of the reference's
PaillierSharedKey / biprimality class-methods driven through a test double of the engine, checked
against the golden vectors recorded from the reference."""

from __future__ import annotations

import random
import re
from pathlib import Path

import numpy as np
import pytest

from conftest import unhex
from fake_engine import FakeEngine
from oracle import oracle

ROOT = Path(__file__).resolve().parent.parent


# ------------------------------------------------------------------ C ABI
def test_library_builds_loads_and_exports_every_declared_symbol():
    from protocols.distributed_keygen_amd import _lib, build

    build.build()
    lib = _lib.load()
    header = (ROOT / "include" / "mxpaillier.h").read_text()
    declared = set(re.findall(r"\b(mx_[a-z_0-9]+)\s*\(", header))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name)
    # ABI 3.0: mx_powmod_nsquare_run gained wavefronts_per_group, mx_nsquare_plan.geometries; 3.1: mx_nsquare_launch_timesliced;
    # 3.2: mx_nsquare_launch_instance, MX_KNOB_N2_FRIENDLY_1W; 3.3: limbs_per_lane 3 for the generic kernel,
    # mx_nsquare_launch_split, MX_KNOB_GENERIC_LATENCY / MX_KNOB_N2_SPLIT
    # 4.0: mx_set_limbs_per_lane removed (no process-wide launch-shape state), knobs atomic, mx_powmod_nsquare_prepare_ex
    # with MX_PLAN_FIXED_WINDOW; 4.3: wavefronts_per_group 4 (the five-wavefront latency form), MX_KNOB_N2_BIPAIR;
    # 4.4: mx_nsquare_latency_form
    assert lib.mx_version() == 404
    assert lib.mx_error_string(-3).decode().startswith("modulus")


def test_geometry_query():
    import ctypes

    from protocols.distributed_keygen_amd import _lib

    lib = _lib.lib()
    k, l, w, b = (ctypes.c_int() for _ in range(4))
    for bits, want_k in ((136, 1), (1028, 4), (2053, 8), (4106, 16), (8206, 32), (16700, 64)):
        assert lib.mx_geometry(bits, k, l, w, b) == 0
        assert (k.value, l.value, w.value) == (want_k, 9, 29)
        assert w.value * l.value * b.value >= bits + 4 and b.value <= k.value
    assert lib.mx_geometry(16701, k, l, w, b) == -2
    # the pair kernel picks its launch shape per launch (tests/test_instances.py pins the crossovers)
    for bits, batch, want in ((2051, 100, (32, 3)), (2051, 3000, (8, 9)), (2051, 7000, (4, 18)), (4100, 100, (64, 3))):
        assert lib.mx_nsquare_geometry(bits, batch, k, l, w, b) == 0
        assert (k.value, l.value) == want, (bits, batch)
        assert w.value * l.value * b.value >= bits + 4
    assert lib.mx_nsquare_geometry(2051, 0, k, l, w, b) == -1
    # kernel timing hooks: nothing launched, nothing recorded
    total, launches = ctypes.c_double(-1.0), ctypes.c_int(-1)
    assert lib.mx_profile(1) == 0 and lib.mx_profile_collect(total, launches) == 0 and lib.mx_profile(0) == 0
    assert (total.value, launches.value) == (0.0, 0)
    assert lib.mx_profile_collect(None, None) == -1
    assert lib.mx_powmod_workspace_bytes(129, 132, 10000, 1) > 0
    assert lib.mx_powmod_workspace_bytes(0, 1, 1, 1) == -1


def test_engine_refuses_to_run_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from protocols.distributed_keygen_amd import Engine

    with pytest.raises(RuntimeError):
        Engine()


# ------------------------------------------------------------------ limbs
def test_pack_unpack_roundtrip():
    from protocols.distributed_keygen_amd import limbs

    rng = random.Random(3)
    vals = [0, 1, (1 << 32) - 1, 1 << 32, (1 << 4128) - 1] + [rng.getrandbits(4100) for _ in range(20)]
    rows = limbs.pack(vals, 129)
    assert rows.shape == (25, 129) and rows.dtype == np.uint32
    assert limbs.unpack(rows) == vals
    assert rows[3, 0] == 0 and rows[3, 1] == 1
    with pytest.raises(ValueError):
        limbs.pack([1 << 64], 2)
    with pytest.raises(ValueError):
        limbs.pack([-1], 2)
    assert limbs.limbs_for(1) == 1 and limbs.limbs_for((1 << 32)) == 2 and limbs.limbs_for(0) == 1


# ------------------------------------------------------------------ PaillierSharedKey mirror
def _keys(grp, engine):
    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, ShareView

    n = unhex(grp["n"])
    return {
        int(i): GpuPaillierSharedKey(
            n=n, t=grp["t"], player_id=int(i),
            share=ShareView({int(i): unhex(s)}, grp["degree"], unhex(grp["n_fac"])),
            theta=unhex(grp["theta"]), engine=engine,
        )
        for i, s in grp["shares"].items()
    }


def test_shared_key_mirror_matches_reference_outputs(golden_ref_keys, golden_decrypt_synth):
    from protocols.distributed_keygen_amd.shared_key import PlainCiphertext

    eng = FakeEngine()
    for src in (golden_ref_keys, golden_decrypt_synth):
        for name, grp in src.items():
            keys = _keys(grp, eng)
            n = unhex(grp["n"])
            assert keys[1].theta_inv == unhex(grp["theta_inv"]) and keys[1].n_square == n * n
            cts = [PlainCiphertext(unhex(c["c"]), n) for c in grp["cases"]]
            if "corrupt" not in name:
                for i, key in keys.items():
                    got = key.partial_decrypt_batch(cts)
                    assert got == [unhex(c["partials"][str(i)]) for c in grp["cases"]], (name, i)
                    assert key.partial_decrypt(cts[0]) == got[0]
                assert all(not c.fresh for c in cts)          # get_value() was used (PSK:69)
            dicts = [{int(i): unhex(v) for i, v in c["partials"].items()} for c in grp["cases"]]
            if any(c["error"] for c in grp["cases"]):
                with pytest.raises(ValueError, match="not divisible by N"):
                    keys[1].decrypt_batch(dicts)
            else:
                assert keys[1].decrypt_batch(dicts) == [unhex(c["m"]) for c in grp["cases"]]
                assert keys[1].decrypt(dicts[0]) == unhex(grp["cases"][0]["m"])
    # one launch per batch, not one per ciphertext
    assert all(count >= 1 for _, count in eng.calls)


def test_shared_key_mirror_error_behaviour(golden_decrypt_synth):
    from protocols.distributed_keygen_amd.shared_key import PlainCiphertext

    grp = golden_decrypt_synth["k128_n3_t1"]
    keys = _keys(grp, FakeEngine())
    n = unhex(grp["n"])
    with pytest.raises(TypeError):                      # PSK:62-65
        keys[1].partial_decrypt(12345)
    with pytest.raises(ValueError, match="different key"):   # PSK:67-68
        keys[1].partial_decrypt(PlainCiphertext(5, n + 2))
    d = {int(i): unhex(v) for i, v in grp["cases"][0]["partials"].items()}
    del d[3]
    with pytest.raises(KeyError):                       # PSK:108-110
        keys[1].decrypt(d)
    assert keys[1].partial_decrypt_batch([]) == [] and keys[1].decrypt_batch([]) == []
    assert keys[1] == keys[1]
    with pytest.raises(TypeError):
        keys[1] == 5  # noqa: B015


# ------------------------------------------------------------------ keygen mirrors
def test_biprime_mirrors_match_reference_outputs(golden_biprime):
    from protocols.distributed_keygen_amd import biprime

    eng = FakeEngine()
    cands = golden_biprime["candidates"]
    for npar in (3, 5):
        sel = [c for c in cands if c["n_parties"] == npar]
        mods = [unhex(c["modulus"]) for c in sel]
        gs = [[unhex(g) for g in c["g_values"]] for c in sel]
        nbips = {c["correct_param_biprime"] for c in sel}
        for nbip in nbips:
            idx = [k for k, c in enumerate(sel) if c["correct_param_biprime"] == nbip]
            v_all = [dict() for _ in idx]
            for i in range(1, npar + 1):
                got = biprime.biprime_test_v_calculation_batch(
                    [gs[k] for k in idx], i, [mods[k] for k in idx],
                    [unhex(sel[k]["p_parts"][i - 1]) for k in idx], [unhex(sel[k]["q_parts"][i - 1]) for k in idx],
                    nbip, engine=eng,
                )
                assert got == [[unhex(x) for x in sel[k]["v"][str(i)]] for k in idx]
                for slot, g in zip(v_all, got):
                    slot[i] = g
            for k, vd in zip(idx, v_all):
                want = sel[k]["verdict"]
                if want == "KeyError":
                    with pytest.raises(KeyError):
                        biprime.biprime_test_with_v_i(vd, mods[k], nbip, engine=eng)
                else:
                    assert biprime.biprime_test_with_v_i(vd, mods[k], nbip, engine=eng) is want, sel[k]["label"]
            ok = [k for k in idx if sel[k]["verdict"] != "KeyError"]
            got = biprime.biprime_test_with_v_i_batch([v_all[idx.index(k)] for k in ok], [mods[k] for k in ok], nbip, engine=eng)
            assert got == [sel[k]["verdict"] for k in ok]


def test_sieve_mirror(golden_biprime):
    from protocols.distributed_keygen_amd import biprime

    eng = FakeEngine()
    for block in golden_biprime["sieve"]:
        primes = oracle.small_prime_list(block["prime_threshold"])
        mods = [unhex(c["modulus"]) for c in block["cases"]]
        assert biprime.small_prime_divisors_test_batch(primes, mods, engine=eng) == [c["has_small_divisor"] for c in block["cases"]]
    assert biprime.small_prime_divisors_test_batch([], [15], engine=eng) == [False]
    assert biprime.small_prime_divisors_test([3, 5], 35, engine=eng) is True


# ------------------------------------------------------------------ wire / disk codec
def test_codec_roundtrip_and_rows():
    from protocols.distributed_keygen_amd import codec, limbs

    rng = random.Random(21)
    vals = [0, 1, 127, 128, 255, 256, (1 << 4102) - 1] + [rng.getrandbits(4100) for _ in range(10)]
    wire = [codec.encode_int(v) for v in vals]
    assert [codec.decode_int(w) for w in wire] == vals
    assert codec.decode_int(codec.encode_int(-5)) == -5
    rows = codec.rows_from_wire(wire, 129)
    assert limbs.unpack(rows) == vals
    assert limbs.unpack(codec.rows_from_wire(vals, 129)) == vals          # plain ints too
    assert [codec.decode_int(w) for w in codec.rows_to_wire(rows)] == vals
    with pytest.raises(ValueError):
        codec.rows_from_wire([codec.encode_int(-1)], 4)
    with pytest.raises(ValueError):
        codec.rows_from_wire([codec.encode_int(1 << 200)], 4)


def test_load_stored_reference_keys_and_decrypt(golden_ref_keys):
    """The reference's own stored test keys (tests/golden/ref_keys/*.obj) load into the mirror and
    decrypt the recorded ciphertexts."""
    from protocols.distributed_keygen_amd import codec
    from protocols.distributed_keygen_amd.shared_key import PlainCiphertext

    eng = FakeEngine()
    keydir = ROOT / "tests" / "golden" / "ref_keys"
    for name, grp in golden_ref_keys.items():
        keys = {}
        for fn in grp["files"]:
            key, meta = codec.load_stored_key((keydir / fn).read_bytes(), engine=eng)
            assert meta["corruption_threshold"] == grp["t"] and key.n == unhex(grp["n"])
            assert key.share.shares[key.player_id] == unhex(grp["shares"][str(key.player_id)])
            keys[key.player_id] = key
        assert sorted(keys) == list(range(1, grp["n_parties"] + 1))
        cts = [PlainCiphertext(unhex(c["c"]), keys[1].n) for c in grp["cases"]]
        partials = {i: k.partial_decrypt_batch(cts) for i, k in keys.items()}
        dicts = [{i: partials[i][e] for i in keys} for e in range(len(cts))]
        assert keys[1].decrypt_batch(dicts) == [unhex(m) for m in grp["plaintexts"]]


# ------------------------------------------------------------------ C ABI argument validation (returns before any HIP call)
def test_c_abi_status_codes_without_gpu():
    import ctypes

    from protocols.distributed_keygen_amd import _lib, limbs

    lib = _lib.lib()
    fake = ctypes.c_void_p(0x1000)          # never dereferenced: every case below fails validation first
    mod = limbs.pack_one((1 << 200) + 235, 8)
    even = limbs.pack_one(1 << 200, 8)
    exp = limbs.pack_one(65537, 1)
    p = lambda a: a.ctypes.data  # noqa: E731
    assert lib.mx_powmod_shared(None, fake, p(mod), p(exp), 8, 1, 4, fake, 1 << 30, None) == -1          # MX_ERR_ARG
    assert lib.mx_powmod_shared(fake, fake, p(mod), p(exp), 0, 1, 4, fake, 1 << 30, None) == -1
    assert lib.mx_powmod_shared(fake, fake, p(even), p(exp), 8, 1, 4, fake, 1 << 30, None) == -3         # MX_ERR_MODULUS
    assert lib.mx_powmod_shared(fake, fake, p(mod), p(exp), 8, 1, 4, fake, 16, None) == -4               # MX_ERR_WORKSPACE
    huge = limbs.pack_one((1 << 16800) + 1, 526)
    assert lib.mx_powmod_shared(fake, fake, p(huge), p(exp), 526, 1, 4, fake, 1 << 40, None) == -2       # MX_ERR_SIZE
    assert lib.mx_powmod_multi(fake, fake, p(mod), p(exp), 8, 1, 1, 0, fake, 1 << 30, None) == -1
    primes = np.array([3, 5, 2], dtype=np.uint32)
    assert lib.mx_sieve(fake, fake, p(primes), 3, 8, 4, fake, 1 << 30, None) == -1                       # even "prime"
    assert lib.mx_sieve(fake, fake, p(primes), 2, 8, 4, fake, 8, None) == -4
    assert lib.mx_combine(fake, fake, fake, p(even), p(mod), 8, 16, 3, 4, fake, 1 << 30, None) == -3
    assert lib.mx_combine(fake, fake, fake, p(mod), p(mod), 8, 8, 3, 4, fake, 1 << 30, None) == -1       # rows too narrow for N^2
    assert lib.mx_biprime_verdict(fake, fake, p(even), 8, 3, 1, 40, fake, 1 << 30, None) == -3
    assert lib.mx_jacobi(fake, fake, p(even), 8, 1, 4, fake, 1 << 30, None) == -3
    assert lib.mx_jacobi(fake, fake, p(mod), 258, 1, 4, fake, 1 << 30, None) == -2
    assert lib.mx_mulmod_shared(fake, fake, fake, p(even), 8, 4, fake, 1 << 30, None) == -3
    assert not hasattr(lib, "mx_set_limbs_per_lane")                 # ABI 4.0: no process-wide launch-shape setting
    # developer knobs, probes and CU-slice streams: argument checks come before any HIP call
    import ctypes

    assert lib.mx_debug_knob(99, 1) == -1 and lib.mx_debug_knob(1, 65) == -1 and lib.mx_debug_knob(2, -1) == -1
    assert lib.mx_debug_knob(3, 3) == -1 and lib.mx_debug_knob(3, 20) == -1
    assert lib.mx_debug_knob(1, 0) == 0 and lib.mx_debug_knob(2, 0) == 0 and lib.mx_debug_knob(3, 0) == 0
    assert lib.mx_spin(-1, None) == -1 and lib.mx_clock_probe(0, fake, None) == -1 and lib.mx_clock_probe(10, None, None) == -1
    sp = ctypes.c_void_p()
    assert lib.mx_stream_create_cu_slice(4, 4, 0, ctypes.byref(sp)) == -1 and lib.mx_stream_create_cu_slice(0, 0, 0, ctypes.byref(sp)) == -1
    assert lib.mx_stream_create_cu_slice(0, 4, 0, None) == -1 and lib.mx_stream_destroy(None) == -1
    assert lib.mx_sieve_workspace_bytes(65, 302) > 0 and lib.mx_combine_workspace_bytes(65, 129, 3, 10) > 0
    assert lib.mx_verdict_workspace_bytes(65, 3, 10, 40) > 0 and lib.mx_jacobi_workspace_bytes(65, 10) > 0
    assert lib.mx_mulmod_workspace_bytes(129) > 0


def test_public_header_is_plain_c():
    """include/mxpaillier.h is the drop-in boundary: it must compile as C99 (and as C++) on its own."""
    import shutil
    import subprocess

    header = str(ROOT / "include" / "mxpaillier.h")
    for compiler, lang in (("gcc", ["-x", "c", "-std=c99"]), ("g++", ["-x", "c++"])):
        if shutil.which(compiler) is None:
            pytest.skip(f"{compiler} not installed")
        subprocess.run([compiler, "-fsyntax-only", "-Wall", "-Werror", *lang, header], check=True)


def test_decrypt_columns_matches_the_per_ciphertext_loop(golden_decrypt_synth):
    """GpuPaillierSharedKey.decrypt_columns (players' columns, as the patched _decrypt_sequence_raw
    collects them) == [decrypt(dict) for every ciphertext], incl. wire-form integers, un-reduced and
    negative values (reduced like mult_list(..., n_square) does), KeyError and ValueError."""
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import codec
    from protocols.distributed_keygen_amd.shared_key import GpuPaillierSharedKey, ShareView

    grp = golden_decrypt_synth["k128_n3_t1"]
    n = unhex(grp["n"])
    n2 = n * n
    share = ShareView({int(i): unhex(s) for i, s in grp["shares"].items()}, grp["degree"], unhex(grp["n_fac"]))
    key = GpuPaillierSharedKey(n, grp["t"], 1, share, unhex(grp["theta"]), engine=FakeEngine())
    cases = grp["cases"]
    cols = {i: [unhex(c["partials"][str(i)]) for c in cases] for i in (1, 2, 3)}
    want = [unhex(c["m"]) for c in cases]
    assert key.decrypt_columns(cols, len(cases)) == want
    assert key.decrypt_columns(cols, 2) == want[:2] and key.decrypt_columns(cols, 0) == []
    # wire form, un-reduced and negative representatives of the same residues
    weird = dict(cols)
    weird[2] = [codec.encode_int(v) for v in cols[2]]
    weird[3] = [v + n2 if k % 2 else v - 3 * n2 for k, v in enumerate(cols[3])]
    assert key.decrypt_columns(weird, len(cases)) == want
    with pytest.raises(KeyError):
        key.decrypt_columns({1: cols[1], 3: cols[3]}, len(cases))
    with pytest.raises(KeyError):
        key.decrypt_columns({1: cols[1], 2: cols[2][:1], 3: cols[3]}, len(cases))
    bad = dict(cols)
    bad[2] = [v + 1 for v in cols[2]]
    with pytest.raises(ValueError):
        key.decrypt_columns(bad, len(cases))


def test_rows_from_wire_fast_path_and_reduction():
    from protocols.distributed_keygen_amd import codec, limbs

    vals = [0, 1, (1 << 200) - 1, 12345678901234567890]
    rows = codec.rows_from_wire(vals, 8)
    assert limbs.unpack(rows) == vals
    mixed = [codec.encode_int(vals[2]), vals[3], codec.encode_int(0)]
    assert limbs.unpack(codec.rows_from_wire(mixed, 8)) == [vals[2], vals[3], 0]
    with pytest.raises(ValueError):
        codec.rows_from_wire([-1], 8)
    with pytest.raises(ValueError):
        codec.rows_from_wire([1 << 300], 8)
    m = (1 << 190) + 7
    assert limbs.unpack(codec.rows_from_wire([-1, m + 5, codec.encode_int(-2), codec.encode_int(1 << 300)], 8, modulus=m)) == [
        m - 1, 5, m - 2, (1 << 300) % m]
    assert limbs.unpack(codec.rows_from_wire(codec.rows_to_wire(rows), 8)) == vals
    # values that fit the row width but are not canonical residues are reduced too: all-int
    # fast path, wire-form entries, mixed lists; without a modulus they pass through unchanged
    assert limbs.unpack(codec.rows_from_wire([m + 5, 3, m, m - 1, 0], 8, modulus=m)) == [5, 3, 0, m - 1, 0]
    assert limbs.unpack(codec.rows_from_wire([codec.encode_int(m + 5)], 8, modulus=m)) == [5]
    assert limbs.unpack(codec.rows_from_wire([codec.encode_int(m + 5), 3, codec.encode_int(m), m + 9], 8, modulus=m)) == [5, 3, 0, 9]
    assert limbs.unpack(codec.rows_from_wire([m + 5], 8)) == [m + 5]
    assert limbs.unpack(codec.rows_from_wire([(1 << 192) - 1], 6, modulus=m)) == [((1 << 192) - 1) % m]
    assert limbs.unpack(codec.rows_from_wire([5], 4, modulus=m)) == [5]            # modulus wider than the rows


def test_bulk_residue_packing():
    """limbs.pack_reduced / reduce_rows: the int-level paths reduce whole columns at once (the word holding a
    modulus' top bit decides for almost every row) instead of one Python `%` per element."""
    import random

    from protocols.distributed_keygen_amd import limbs

    rng = random.Random(8)
    for bits in (33, 64, 65, 200, 2051):
        m = rng.getrandbits(bits) | (1 << (bits - 1)) | 1
        width = limbs.limbs_for(m) + (1 if bits == 200 else 0)
        top = 1 << (32 * width)
        vals = [0, 1, m - 1, m, m + 1, 2 * m - 1, top - 1] + [rng.randrange(m) for _ in range(50)] + [rng.getrandbits(32 * width) for _ in range(20)]
        assert limbs.unpack(limbs.pack_reduced(vals, width, m)) == [v % m for v in vals], bits
        odd = vals + [-5, top << 3]                      # negative / too wide: the per-element path
        assert limbs.unpack(limbs.pack_reduced(odd, width, m)) == [v % m for v in odd], bits
    mods = [rng.getrandbits(130) | (1 << 129) | 1 for _ in range(4)]
    vals = [rng.getrandbits(160) for _ in range(4 * 7)]
    assert limbs.unpack(limbs.pack_reduced(vals, 5, mods)) == [v % mods[k // 7] for k, v in enumerate(vals)]
    assert limbs.pack_reduced([], 3, 7).shape == (0, 3)
    wide = 1 << 200                                       # modulus wider than the rows: nothing to reduce
    assert limbs.unpack(limbs.pack_reduced([5, (1 << 96) - 1], 3, wide)) == [5, (1 << 96) - 1]


def test_c_codec_matches_int_to_bytes():
    import random

    from protocols.distributed_keygen_amd import limbs

    assert limbs._codec() is not None, "the C codec (csrc/mx_pycodec.c) must be built with the library"
    rng = random.Random(5)
    vals = [0, 1, (1 << 4128) - 1] + [rng.getrandbits(rng.randrange(1, 4128)) for _ in range(200)]
    rows = limbs.pack(vals, 129)
    assert [int.from_bytes(r.tobytes(), "little") for r in rows] == vals
    assert limbs.unpack(rows) == vals
    out = np.zeros((len(vals) + 3, 129), dtype="<u4")
    limbs.pack_into(vals, 129, out, 3)
    assert limbs.unpack(out[3:]) == vals and not out[:3].any()
    for bad in ([-5], [1 << 4128]):
        with pytest.raises(ValueError):
            limbs.pack(bad, 129)
    with pytest.raises(ValueError):
        limbs.pack_into(vals, 129, np.zeros((10, 129), dtype="<u4"), 0)      # buffer too small
    # the threaded paths (many elements, every thread count) agree with the single-threaded one and with int.to_bytes
    codec = limbs._codec()
    assert codec.DIRECT_DIGITS == 1
    many = [rng.getrandbits(rng.choice([1, 29, 30, 31, 59, 60, 61, 2053, 4100])) for _ in range(40000)] + [0, (1 << 4128) - 1, True]
    ref = np.frombuffer(b"".join(int(v).to_bytes(516, "little") for v in many), dtype="<u4").reshape(-1, 129)
    for threads in (1, 2, 7, 16, 0):
        prev = codec.set_threads(threads)
        try:
            got = limbs.pack(many, 129)
            assert (got == ref).all(), threads
            back = limbs.unpack(got)
            assert back == [int(v) for v in many] and all(type(v) is int for v in back), threads
        finally:
            codec.set_threads(prev)
    with pytest.raises(TypeError):
        codec.pack_into(many[:5000] + [1.5], 129, np.zeros((5001, 129), dtype="<u4"), 0)     # a failure inside a worker's slice
    with pytest.raises(ValueError):
        codec.pack_into(many[:5000] + [-1], 129, np.zeros((5001, 129), dtype="<u4"), 0)
    # rows_ge: the one-pass residue check behind limbs.reduce_rows
    mods = [(1 << 100) + 7, (1 << 64) - 1, 12345]
    vals3 = [5, (1 << 100) + 7, (1 << 100) + 6, (1 << 64) - 1, (1 << 64) - 2, 1 << 90, 12345, 12344, 0]
    assert codec.rows_ge(limbs.pack(vals3, 4), 4, limbs.pack(mods, 4), 3) == [1, 3, 5, 6]
    assert limbs.unpack(limbs.reduce_rows(limbs.pack(vals3, 4), mods)) == [v % mods[k // 3] for k, v in enumerate(vals3)]


def test_shamir_mirror_matches_reference_vectors(golden_reconstruct):
    from fake_engine import FakeEngine
    from protocols.distributed_keygen_amd import shamir

    eng = FakeEngine()
    for label, grp in golden_reconstruct.items():
        prime, degree = unhex(grp["prime"]), grp["degree"]
        shares = {int(i): {k: [unhex(v) for v in vals] for k, vals in d.items()} for i, d in grp["shares"].items()}
        for i, d in shares.items():
            assert shamir.mul_add_shares_batch(d["p"], d["q"], d["zero"], prime, eng) == d["n"]
        want = [unhex(m) for m in grp["moduli"]]
        assert shamir.reconstruct_batch({i: d["n"] for i, d in shares.items()}, prime, degree, eng) == want
        # any degree+1 shares determine the same polynomial: dropping surplus parties changes nothing
        some = dict(list({i: d["n"] for i, d in shares.items()}.items())[-(degree + 1):])
        assert shamir.reconstruct_batch(some, prime, degree, eng) == want
        with pytest.raises(ValueError):
            shamir.reconstruct_batch(dict(list(some.items())[1:]), prime, degree, eng)
        # explicit interpolation points (the insertion order of the reference's shares dictionary, as
        # patch.compute_modulus passes them): same moduli from consistent shares; with one party's shares
        # corrupted the result depends on whether that party is among the points — as in the reference
        allp = {i: d["n"] for i, d in shares.items()}
        order = list(reversed(sorted(allp)))
        assert shamir.reconstruct_batch(allp, prime, degree, eng, points=order) == want
        if len(allp) > degree + 1:
            bad = dict(allp)
            bad[order[-1]] = [(v + 1) % prime for v in bad[order[-1]]]
            assert shamir.reconstruct_batch(bad, prime, degree, eng, points=order) == want       # corrupted party not used
            assert shamir.reconstruct_batch(bad, prime, degree, eng, points=sorted(allp)) != want
        with pytest.raises(KeyError):
            shamir.reconstruct_batch(allp, prime, degree, eng, points=[99] + order)
    assert shamir.reconstruct_batch({1: [], 2: [], 3: []}, 101, 2, eng) == []
    assert shamir.lagrange_coefficients_at_zero([1, 2, 3], 101) == [3, 98, 1]


def _posdivsteps_jacobi_model(x: int, n: int, steps: int = 30) -> int:
    """Python model of csrc/mx_jacobi.hpp: batches of `steps` all-positive divsteps decided from the
    low 64 bits, a 2x2 matrix per batch applied to the full operands, sign tracked on the low bits."""
    m64 = (1 << 64) - 1
    if n == 1:
        return 1
    if x == 0 or n % 2 == 0:
        return 0
    f, g, eta, jac = n, x, -1, 0
    for _ in range((n.bit_length() + 31) // 32 * 32 * 4 // steps + 8):
        u, v, q, r = 1, 0, 0, 1
        fl, gl, i = f & m64, g & m64, steps
        while True:
            t = (gl | (m64 << i)) & m64
            zeros = (t & -t).bit_length() - 1
            gl >>= zeros
            u <<= zeros
            v <<= zeros
            eta -= zeros
            i -= zeros
            jac ^= zeros & ((fl >> 1) ^ (fl >> 2)) & 1
            if i == 0:
                break
            if eta < 0:
                eta = -eta
                fl, gl, u, q, v, r = gl, fl, q, u, r, v
                jac ^= ((fl & gl) >> 1) & 1
                mask = (m64 >> (64 - min(eta + 1, i))) & 63
                w = (fl * gl * (fl * fl - 2)) & mask
            else:
                mask = (m64 >> (64 - min(eta + 1, i))) & 15
                w = (-(fl + (((fl + 1) & 4) << 1)) * gl) & mask
            gl = (gl + fl * w) & m64
            q += u * w
            r += v * w
        assert max(u, v, q, r) <= 1 << steps
        f, g = (u * f + v * g) >> steps, (q * f + r * g) >> steps
        assert f <= n and g <= n
        if f == 1:
            return 1 - 2 * (jac & 1)
        if f == g:
            return 0
    raise AssertionError("no convergence")


def test_divstep_jacobi_model_matches_sympy():
    """The algorithm of the Jacobi kernel (batched all-positive divsteps), validated against sympy."""
    import random

    import sympy

    rng = random.Random(2053)
    for t in range(4000):
        bits = rng.choice([2, 3, 5, 8, 16, 31, 32, 33, 61, 64, 65, 131, 520, 1028])
        n = rng.getrandbits(bits) | 1
        x = rng.randrange(n) if n > 1 else 0
        if t % 6 == 0:                      # a common factor: the symbol is 0
            d = rng.choice([3, 5, 7, 9, 15, 21, 2**31 - 1])
            n, x = n * d, (x * d) % (n * d)
        assert _posdivsteps_jacobi_model(x, n) == sympy.jacobi_symbol(x, n), (x, n)
    for n in (3, 5, 7, 9, 2**64 - 1, 2**64 + 13):
        for x in (0, 1, 2, n - 1, n - 2, (n + 1) // 2):
            assert _posdivsteps_jacobi_model(x % n, n) == sympy.jacobi_symbol(x % n, n)


class _IntLike:
    """an integer scalar that is not an int (what gmpy2.mpz / numpy integers are to isinstance(x, int))"""

    def __init__(self, v):
        self.v = int(v)

    def __index__(self):
        return self.v

    def __int__(self):
        return self.v


def test_int_like_moduli_are_one_modulus_not_a_sequence():
    """reduce_rows / pack_reduced told one modulus from a list with isinstance(moduli, int), so a
    numpy.int64 or gmpy2.mpz modulus was iterated (TypeError)."""
    from protocols.distributed_keygen_amd import limbs as L

    m = (1 << 61) - 1
    vals = [5, m + 3, 2 * m + 1, (1 << 64) + 7]
    want = L.pack([v % m for v in vals], 3)
    for mod in (m, np.int64(m), np.uint64(m), _IntLike(m)):
        assert (L.pack_reduced(vals, 3, mod) == want).all(), type(mod)
        assert (L.reduce_rows(L.pack(vals, 3), mod) == want).all(), type(mod)
        assert (L.pack_reduced(vals + [-1], 3, mod)[:4] == want).all(), type(mod)      # the per-element path (negative value)
    # a sequence of int-like moduli, one per group of consecutive rows
    mods = [np.int64(m), _IntLike(97)]
    got = L.pack_reduced([m + 1, m + 2, 100, 200], 3, mods)
    assert L.unpack(got) == [1, 2, 3, 6]


def test_engine_entry_points_coerce_int_like_operands():
    from protocols.distributed_keygen_amd import engine

    seen = {}

    @engine._int_args
    def entry(self, bases: Sequence[int], exp: int, mods: Sequence[int], mod: int = 7, other=None):
        seen.update(exp=exp, mods=mods, mod=mod, bases=bases, other=other)

    entry(None, [_IntLike(3)], np.int64(5), [np.int64(9), _IntLike(11)], mod=_IntLike(13), other=np.int64(1))
    assert type(seen["exp"]) is int and seen["exp"] == 5
    assert seen["mods"] == [9, 11] and all(type(x) is int for x in seen["mods"])
    assert type(seen["mod"]) is int and seen["mod"] == 13
    assert isinstance(seen["other"], np.int64) and isinstance(seen["bases"][0], _IntLike)      # untouched
    # every int-level entry point of the Engine is wrapped
    for name in ("powmod_batch", "powmod_batch_multi", "powmod_nsquare_batch", "modinv_batch", "combine_batch", "sieve_batch",
                 "jacobi_batch", "biprime_v_batch", "biprime_verdict_batch", "shamir_lincomb_batch", "encrypt_batch"):
        assert hasattr(getattr(engine.Engine, name), "__wrapped__"), name


def test_biprime_round_keeps_state_between_steps_and_never_trusts_a_changed_column():
    """biprime.BiprimeRound through the test double's device-resident forms: the survivors' moduli handle goes from the
    sieve to the v-calculation and the verdicts, this party's v handle stands in for its column only while the values
    handed to verdicts() equal the computed ones, and the results equal the list-level functions either way."""
    import random

    import sympy

    from protocols.distributed_keygen_amd import biprime

    rng = random.Random(11)
    prime = int(sympy.nextprime(1 << 150))
    degree, points = 2, [1, 2, 3]
    moduli = [(rng.getrandbits(62) | 1) * (rng.getrandbits(62) | 1) for _ in range(400)]
    cols = {i: [] for i in points}
    for m in moduli:
        co = [m] + [rng.randrange(prime) for _ in range(degree)]
        for i in points:
            cols[i].append(sum(c * i ** k for k, c in enumerate(co)) % prime)
    primes = [int(q) for q in sympy.primerange(3, 150)]
    eng = FakeEngine()
    rnd = biprime.BiprimeRound(eng)
    surviving = rnd.reconstruct_and_sieve(cols, prime, degree, primes, points=points)
    assert surviving == {k: m for k, m in enumerate(moduli) if not oracle.small_prime_divisors_test(primes, m)}
    assert rnd.survivors == sorted(surviving) and rnd.moduli == [surviving[k] for k in rnd.survivors] and len(rnd.moduli) >= 10
    g = [[rng.randrange(m) for _ in range(20)] for m in rnd.moduli]
    g[3] = g[3][:2]
    ps = [rng.getrandbits(50) for _ in rnd.moduli]
    qs = [rng.getrandbits(50) for _ in rnd.moduli]
    v2 = rnd.v_calculation(g, 2, ps, qs, 5)
    assert v2 == biprime.biprime_test_v_calculation_batch(g, 2, rnd.moduli, ps, qs, 5, FakeEngine())
    others = {i: [[rng.randrange(m) for _ in range(5)] for m in rnd.moduli] for i in (1, 3)}
    v_by = [{1: others[1][c], 2: list(v2[c]), 3: others[3][c]} for c in range(len(rnd.moduli))]
    want = biprime.biprime_test_with_v_i_batch(v_by, rnd.moduli, 5, FakeEngine(), errors="return")
    eng.calls.clear()
    got = rnd.verdicts(v_by, 5, errors="return")
    assert [repr(x) for x in got] == [repr(x) for x in want]
    assert ("own_column_from_device", len(rnd.moduli)) in eng.calls            # equal values: the kept rows stood in
    v_by[0][2] = [(v_by[0][2][0] + 1) % rnd.moduli[0]] + v_by[0][2][1:]
    eng.calls.clear()
    got2 = rnd.verdicts(v_by, 5, errors="return")
    assert not any(c[0] == "own_column_from_device" for c in eng.calls)        # a changed column is packed like any other
    assert [repr(x) for x in got2] == [repr(x) for x in biprime.biprime_test_with_v_i_batch(v_by, rnd.moduli, 5, FakeEngine(), errors="return")]


def test_sieve_columns_never_overflow_64_bits_with_the_hosts_chunk():
    """csrc/mx_sieve.hpp / mx_sieve in mx_capi.hip restated with Python ints: with chunk = (2^32 - 2) // top - 1 limbs
    between two folds, a 64-bit column never reaches 2^64 — worst case: every candidate limb 2^32 - 1 and every table
    entry l - 1 — and the folded column keeps the residue."""
    import random

    rng = random.Random(5)
    M64 = 1 << 64
    for top in (3, 2000, (1 << 21) - 9, (1 << 21) + 17, (1 << 26) + 15, (1 << 30) + 3, (1 << 31) - 1):
        chunk = (0xFFFFFFFE // top) - 1
        assert chunk >= 1
        for limbs in (1, 2, 3, 64, 257, 1024):
            step = min(chunk, limbs)
            for l in {top, 3, max(3, top // 2) | 1}:
                pw = [pow(2, 32 * j, l) for j in range(limbs)]
                for worst in (True, False):
                    cand = [0xFFFFFFFF] * limbs if worst else [rng.getrandbits(32) for _ in range(limbs)]
                    table = [l - 1] * limbs if worst else pw          # the bound must hold for ANY entry below l
                    acc, peak = 0, 0
                    for j0 in range(0, limbs, step):
                        j1 = min(j0 + step, limbs)
                        for j in range(j0, j1):
                            acc += cand[j] * table[j]
                            peak = max(peak, acc)
                        if j1 < limbs:
                            acc = (acc & 0xFFFFFFFF) + (acc >> 32) * (table[1] if worst else pw[1])
                            peak = max(peak, acc)
                    assert peak < M64, (top, limbs, l, worst)
                    if not worst:
                        n = sum(c << (32 * j) for j, c in enumerate(cand))
                        assert acc % l == n % l, (top, limbs, l)


def test_latency_geometry_is_passed_to_generic_launches_only_where_it_exists():
    """Engine._lpl_generic — 3 limbs per lane for generic moduli up to 5533 bits, automatic beyond."""
    from protocols.distributed_keygen_amd.engine import Engine

    e = Engine.__new__(Engine)          # no GPU: only the setting and the rule
    e._lpl = 3
    assert e._lpl_generic(2053) == 3 and e._lpl_generic(5533) == 3 and e._lpl_generic(5534) == 0 and e._lpl_generic(8200) == 0
    for lpl in (0, 9, 18):
        e._lpl = lpl
        assert e._lpl_generic(8200) == lpl and e._lpl_generic(1029) == lpl


def test_an_evicted_plan_stays_readable_for_launches_in_flight_on_every_stream():
    """VERDICT r05 "weak" 10: Engine.MAX_PLANS keys are kept; the 17th evicts the least recently used plan, whose device
    block goes back to torch's caching allocator.  The allocator reuses a block only behind (a) the stream it was
    allocated on and (b) every stream named with record_stream — so the engine must have named every OTHER stream whose
    launch reads the plan BEFORE that launch, whatever happens to the cache afterwards.  Engine plumbing on a stand-in for
    torch's streams / allocator (no GPU): a plan prepared on stream A and used on B and C is evicted by 16 more keys
    while B's and C's launches are "pending"; the block's wait-set must contain all three streams."""
    from collections import OrderedDict

    from protocols.distributed_keygen_amd.engine import Engine, _Plan

    class Stream:
        def __init__(self, ptr):
            self.cuda_stream = ptr

        def wait_event(self, ev):
            ev.waited_by.append(self.cuda_stream)

    class Event:
        def __init__(self):
            self.waited_by = []

        def query(self):
            return False          # the uploads are "still running"

    class Block:
        """What the caching allocator knows about a block: it may be reused behind these streams' work."""

        def __init__(self, alloc_stream):
            self.wait_set = {alloc_stream}

        def record_stream(self, stream):
            self.wait_set.add(stream.cuda_stream)

    class Cuda:
        current = Stream(0xA)

        @staticmethod
        def current_stream(device=None):
            return Cuda.current

    class Torch:
        cuda = Cuda

    eng = Engine.__new__(Engine)
    eng.torch, eng.device = Torch, None
    cache = OrderedDict()
    first = _Plan(desc=object(), block=Block(0xA), stream_ptr=0xA, ready=Event())
    eng._cache_plan(cache, "key0", first)
    for ptr in (0xB, 0xC):                       # launches on two other streams read the plan ...
        Cuda.current = Stream(ptr)
        eng._use_plan(first)
    assert first.ready.waited_by == [0xB, 0xC]   # ... behind its uploads
    Cuda.current = Stream(0xA)
    eng._use_plan(first)                         # the preparing stream itself needs neither
    assert first.ready.waited_by == [0xB, 0xC]
    for k in range(1, Engine.MAX_PLANS + 1):     # ... and stay "pending" while 16 more keys arrive
        eng._cache_plan(cache, f"key{k}", _Plan(object(), Block(0xA), 0xA, None))
    assert "key0" not in cache and len(cache) == Engine.MAX_PLANS
    assert first.block.wait_set == {0xA, 0xB, 0xC}          # the allocator will not hand the block out before all three are done
    # least RECENTLY USED goes first: a plan that was just looked up again survives the next key
    cache.move_to_end("key1")                    # what nsquare_plan / combine_plan do on a hit
    eng._cache_plan(cache, "key17", _Plan(object(), Block(0xA), 0xA, None))
    assert "key1" in cache and "key2" not in cache


def test_nested_codec_calls_match_the_flat_ones():
    """limbs.pack_nested_into / unpack_groups (csrc/mx_pycodec.c): one list per candidate in, one list per candidate out
    — the same rows and ints as flattening / slicing by hand, for ragged lists, zero padding, empty groups, big and zero
    values; errors as the flat calls raise them; the pure-Python fallbacks agree."""
    from protocols.distributed_keygen_amd import limbs as L

    rng = random.Random(606)
    limbs = 9
    lists = [[rng.getrandbits(rng.choice([1, 30, 31, 60, 200, 287])) for _ in range(rng.choice([0, 1, 3, 5, 7]))] for _ in range(2500)]
    lists[3] = [0, (1 << 288) - 1, 1 << 287, 1]
    inner = 5
    want_flat = []
    for vals in lists:
        want_flat += vals[:inner] + [0] * (inner - min(inner, len(vals)))
    out = np.full((len(lists) * inner + 2, limbs), 0xAAAAAAAA, dtype="<u4")
    L.pack_nested_into(lists, inner, limbs, out, 1)
    assert (out[1:-1] == L.pack(want_flat, limbs)).all() and (out[0] == 0xAAAAAAAA).all() and (out[-1] == 0xAAAAAAAA).all()
    tuples = tuple(tuple(v) for v in lists[:50])
    out2 = np.zeros((50 * inner, limbs), dtype="<u4")
    L.pack_nested_into(tuples, inner, limbs, out2)
    assert (out2 == out[1:1 + 50 * inner]).all()
    with pytest.raises(ValueError):
        L.pack_nested_into([[1 << 288]], 1, limbs, np.zeros((1, limbs), dtype="<u4"))
    with pytest.raises(ValueError):
        L.pack_nested_into([[-1]], 1, limbs, np.zeros((1, limbs), dtype="<u4"))
    with pytest.raises(ValueError):
        L.pack_nested_into([[1], [2]], 2, limbs, np.zeros((3, limbs), dtype="<u4"))          # buffer too small
    # groups out
    rows = L.pack(want_flat, limbs)
    counts = [min(inner, len(v)) for v in lists]
    got = L.unpack_groups(rows, counts, inner)
    assert got == [v[:inner] for v in lists] and all(type(x) is int for g in got for x in g)
    assert L.unpack_groups(rows[:0], [], inner) == []
    with pytest.raises(ValueError):
        L.unpack_groups(rows, [inner + 1] + counts[1:], inner)
    with pytest.raises(ValueError):
        L.unpack_groups(rows[: inner * 3], counts[:4], inner)
    # the fallbacks (an interpreter without the C helper)
    saved = L._mxcodec
    try:
        L._mxcodec = None
        L._codec_tried = True if hasattr(L, "_codec_tried") else None
        orig = L._codec
        L._codec = lambda: None
        out3 = np.zeros((len(lists) * inner, limbs), dtype="<u4")
        L.pack_nested_into(lists, inner, limbs, out3)
        assert (out3 == out[1:-1]).all() and L.unpack_groups(rows, counts, inner) == got
    finally:
        L._codec = orig
        L._mxcodec = saved


def test_codec_threads_see_consistent_inputs_and_report_rows_in_order():
    """csrc/mx_pycodec.c after round 6: packing keeps the interpreter lock and borrows the list's own element pointers (no
    per-element reference counting) — the values of an ITERATOR, which exist nowhere but in the temporary list the codec
    builds, must stay alive until the rows are written; rows_ge compares on several threads and still reports ascending
    indices, however many there are."""
    from protocols.distributed_keygen_amd import limbs as L

    limbs = 5
    rng = random.Random(77)
    # inner sequences that are neither lists nor tuples: generators of FRESH ints (no other reference anywhere)
    seeds = [rng.getrandbits(150) for _ in range(3000)]
    gens = [map(lambda k, s=s: s + k, range(4)) for s in seeds]
    out = np.zeros((3000 * 4, limbs), dtype="<u4")
    L.pack_nested_into(gens, 4, limbs, out)
    assert L.unpack(out) == [s + k for s in seeds for k in range(4)]
    # reference counts of the elements are what they were (nothing leaked, nothing dropped)
    import sys

    vals = [rng.getrandbits(159) for _ in range(5000)]
    before = [sys.getrefcount(v) for v in vals[:50]]
    rows = L.pack(vals, limbs)
    L.pack_nested_into([vals[:7], vals[7:9]], 7, limbs, np.zeros((14, limbs), dtype="<u4"))
    assert [sys.getrefcount(v) for v in vals[:50]] == before
    # rows_ge over 40 000 rows with every third row >= its group's modulus
    mods = [(1 << 158) + 2 * k + 1 for k in range(400)]
    group = 100
    big = [(mods[k // group] + (k % 5)) if k % 3 == 0 else rng.randrange(mods[k // group]) for k in range(len(mods) * group)]
    rows = L.pack(big, limbs)
    codec = L._codec()
    if codec is not None:
        assert codec.rows_ge(rows, limbs, L.pack(mods, limbs), group) == [k for k in range(len(big)) if k % 3 == 0]
        assert codec.rows_ge(rows[:0], limbs, L.pack(mods, limbs), group) == []
    reduced = L.reduce_rows(rows.copy(), mods)
    assert L.unpack(reduced) == [v % mods[k // group] for k, v in enumerate(big)]
