"""GPU parity: batched modexp through the C ABI vs the oracle (bit-exact).

Covers every lanes-per-element geometry (K = 1..64), edge operands, ragged batches, and the
golden vectors recorded from the reference (partial decryptions PSK:92, v-values DK:1094/1097).
"""

from __future__ import annotations

import random

import pytest

from conftest import unhex
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from protocols.distributed_keygen_amd import Engine

    return Engine()


def test_lane_primitives_selftest(eng):
    assert eng.selftest_lanes() == 0


@pytest.mark.parametrize(
    "mod_bits,exp_bits,batch",
    [
        (20, 70, 7),        # K=1, tiny
        (136, 200, 65),     # K=1 (N^2 of the key_length=64 fixtures)
        (257, 300, 33),     # K=1, largest
        (258, 300, 33),     # K=2, smallest
        (518, 64, 9),       # K=2
        (1028, 1026, 50),   # K=4  (biprime test, key_length 1024)
        (1040, 100, 17),    # K=4 largest
        (2053, 2051, 24),   # K=8  (biprime test, key_length 2048)
        (4106, 600, 21),    # K=16 (partial decryption modulus, key_length 2048)
        (8206, 200, 5),     # K=32 (key_length 4096)
        (16700, 64, 3),     # K=64 largest supported
    ],
)
def test_powmod_shared_random(eng, mod_bits, exp_bits, batch):
    rng = random.Random(mod_bits * 1000 + exp_bits)
    mod = rng.getrandbits(mod_bits) | (1 << (mod_bits - 1)) | 1
    exp = rng.getrandbits(exp_bits) | (1 << (exp_bits - 1))
    bases = [rng.randrange(mod) for _ in range(batch)]
    bases[0] = 0
    bases[1] = 1
    bases[2] = mod - 1
    got = eng.powmod_batch(bases, exp, mod)
    assert got == [oracle.pow_mod(b, exp, mod) for b in bases]


@pytest.mark.parametrize("exp", [0, 1, 2, 3, 31, 32, 33, (1 << 64) - 1, 1 << 64])
def test_powmod_edge_exponents(eng, exp):
    rng = random.Random(exp % 1000)
    mod = rng.getrandbits(300) | (1 << 299) | 1
    bases = [0, 1, 2, mod - 1, mod - 2] + [rng.randrange(mod) for _ in range(12)]
    assert eng.powmod_batch(bases, exp, mod) == [oracle.pow_mod(b, exp, mod) for b in bases]


def test_powmod_small_moduli(eng):
    for mod in (3, 5, 7, 9, 255, 257, (1 << 29) - 1, (1 << 29) + 1, (1 << 32) - 1, (1 << 32) + 1, (1 << 58) + 1):
        bases = list(range(0, min(mod, 40)))
        assert eng.powmod_batch(bases, 12345, mod) == [oracle.pow_mod(b, 12345, mod) for b in bases], mod


def test_powmod_unreduced_bases_are_reduced_like_pow(eng):
    mod = (1 << 200) + 235
    bases = [mod, mod + 1, 3 * mod + 7, -5, -mod - 2]
    assert eng.powmod_batch(bases, 77, mod) == [pow(b, 77, mod) for b in bases]


def test_powmod_empty_and_errors(eng):
    assert eng.powmod_batch([], 5, 7) == []
    with pytest.raises(ValueError):
        eng.powmod_batch([1, 2], 5, 8)
    with pytest.raises(ValueError):
        eng.powmod_batch([1, 2], -5, 7)


@pytest.mark.parametrize("mod_bits,groups,gsize", [(131, 9, 40), (1028, 6, 40), (2053, 3, 40), (300, 5, 7)])
def test_powmod_multi_random(eng, mod_bits, groups, gsize):
    rng = random.Random(mod_bits + groups)
    mods = [rng.getrandbits(mod_bits - k % 3) | (1 << (mod_bits - k % 3 - 1)) | 1 for k in range(groups)]
    exps = [rng.getrandbits(mod_bits - 2 - (k % 5)) for k in range(groups)]
    exps[0] = 0
    bases = [[rng.randrange(m) for _ in range(gsize if g != 1 else gsize - 3)] for g, m in enumerate(mods)]
    got = eng.powmod_batch_multi(bases, exps, mods)
    want = [[oracle.pow_mod(b, e, m) for b in bs] for bs, e, m in zip(bases, exps, mods)]
    assert got == want


def test_golden_partial_decryptions(eng, golden_decrypt_synth, golden_ref_keys):
    """c^exp_i mod N^2 for every party of every recorded key (reference outputs, PSK:52-93)."""
    for src in (golden_ref_keys, golden_decrypt_synth):
        for name, grp in src.items():
            if "corrupt" in name:
                continue
            n = unhex(grp["n"])
            n2 = n * n
            cs = [unhex(c["c"]) for c in grp["cases"]]
            for i, share in grp["shares"].items():
                exp = oracle.partial_decrypt_exponent(int(i), grp["degree"], unhex(grp["n_fac"]), unhex(share))
                bases = cs if exp >= 0 else [oracle.mod_inv(c, n2) for c in cs]
                got = eng.powmod_batch(bases, abs(exp), n2)
                assert got == [unhex(c["partials"][i]) for c in grp["cases"]], (name, i)


def test_golden_biprime_v_values(eng, golden_biprime):
    """v = g^e mod N for the Jacobi-1 generators of every recorded candidate (DK:1084-1099)."""
    by_shape = {}
    for cand in golden_biprime["candidates"]:
        modulus = unhex(cand["modulus"])
        gs = [unhex(g) for g in cand["g_values"]]
        keep = [g for g in gs if oracle.jacobi_symbol(g, modulus) == 1][: cand["correct_param_biprime"]]
        for i in range(1, cand["n_parties"] + 1):
            e = oracle.biprime_exponent(i, modulus, unhex(cand["p_parts"][i - 1]), unhex(cand["q_parts"][i - 1]))
            want = [unhex(v) for v in cand["v"][str(i)]]
            by_shape.setdefault((modulus.bit_length() // 64, i == 1), []).append((keep, e, modulus, want))
    for items in by_shape.values():
        got = eng.powmod_batch_multi([k for k, _, _, _ in items], [e for _, e, _, _ in items], [m for _, _, m, _ in items])
        assert got == [w for _, _, _, w in items]


# ------------------------------------------------------------------ modulus N^2 through pairs modulo N
@pytest.mark.parametrize("n_bits,exp_bits,batch", [(20, 45, 9), (68, 150, 33), (131, 300, 20), (257, 600, 7),
                                                   (515, 64, 5), (1028, 2100, 6), (2051, 700, 17), (4099, 128, 3)])
def test_powmod_nsquare_random(eng, n_bits, exp_bits, batch):
    rng = random.Random(n_bits * 7 + exp_bits)
    n = rng.getrandbits(n_bits) | (1 << (n_bits - 1)) | 1
    n2 = n * n
    exp = rng.getrandbits(exp_bits) | (1 << (exp_bits - 1))
    bases = [rng.randrange(n2) for _ in range(batch)]
    bases[0], bases[1], bases[2] = 0, 1, n2 - 1
    if batch > 4:
        bases[3], bases[4] = n, n + 1                      # multiples of N / the Paillier generator
    assert eng.powmod_nsquare_batch(bases, exp, n) == [pow(b, exp, n2) for b in bases]


@pytest.mark.parametrize("exp", [0, 1, 2, 3, 255, 256, (1 << 64) + 1])
def test_powmod_nsquare_edge_exponents(eng, exp):
    rng = random.Random(exp % 997)
    n = rng.getrandbits(300) | (1 << 299) | 1
    n2 = n * n
    bases = [0, 1, 2, n - 1, n, n + 1, n2 - 1, n2 - n] + [rng.randrange(n2) for _ in range(9)]
    assert eng.powmod_nsquare_batch(bases, exp, n) == [pow(b, exp, n2) for b in bases]


def test_powmod_nsquare_special_moduli(eng):
    rng = random.Random(12)
    for n in (3, 5, (1 << 29) - 1, (1 << 29) + 1, (1 << 261) - 1, (1 << 2050) + 1, (1 << 2053) - 1):
        n2 = n * n
        bases = [0, 1, 2, n2 - 1, n2 // 2, (1 << (n2.bit_length() - 1)) - 1] + [rng.randrange(n2) for _ in range(6)]
        bases = [b % n2 for b in bases]
        e = rng.getrandbits(90) | 1
        assert eng.powmod_nsquare_batch(bases, e, n) == [pow(b, e, n2) for b in bases], n.bit_length()


def test_powmod_nsquare_golden_partial_decryptions(eng, golden_decrypt_synth, golden_ref_keys):
    for src in (golden_ref_keys, golden_decrypt_synth):
        for name, grp in src.items():
            if "corrupt" in name:
                continue
            n = unhex(grp["n"])
            n2 = n * n
            cs = [unhex(c["c"]) for c in grp["cases"]]
            for i, share in grp["shares"].items():
                exp = oracle.partial_decrypt_exponent(int(i), grp["degree"], unhex(grp["n_fac"]), unhex(share))
                bases = cs if exp >= 0 else [oracle.mod_inv(c, n2) for c in cs]
                assert eng.powmod_nsquare_batch(bases, abs(exp), n) == [unhex(c["partials"][i]) for c in grp["cases"]], (name, i)


def test_kernel_timing_hooks_and_concurrent_streams(eng):
    """mx_profile brackets each modexp kernel with events on the caller's stream; launches issued on
    several streams without any host synchronisation in between (the C ABI never synchronises) all
    complete and are all recorded — what bench.py relies on."""
    import torch

    from protocols.distributed_keygen_amd import Engine, limbs as L

    rng = random.Random(4242)
    n = rng.getrandbits(1027) | (1 << 1026) | 1
    n2 = n * n
    e = rng.getrandbits(300) | 1
    bases = [rng.randrange(n2) for _ in range(64)]
    want = [oracle.pow_mod(b, e, n2) for b in bases]
    rows = eng.to_device(L.pack(bases, L.limbs_for(n2)))
    engines = [eng, Engine(), Engine()]
    streams = [torch.cuda.Stream() for _ in engines]
    torch.cuda.synchronize()
    eng.profile(True)
    outs = []
    for k in range(6):
        with torch.cuda.stream(streams[k % 3]):
            outs.append(engines[k % 3].powmod_nsquare_t(rows, n, e) if k % 2 == 0 else engines[k % 3].powmod_shared_t(rows, n2, e))
    eng.profile(False)
    total_ms, launches = eng.profile_collect()
    torch.cuda.synchronize()
    assert launches == 6 and total_ms > 0
    for out in outs:
        assert L.unpack(eng.to_host(out)) == want
    assert eng.profile_collect() == (0.0, 0)


def test_one_engine_driven_from_several_streams(eng):
    """one Engine used under several torch streams.  Each stream has its own
    workspace inside the engine (the window tables of launches in flight must not overlap) and the
    per-key plan prepared on one stream is ordered before its use on the others.  Eight launches with
    different exponents / keys interleaved over four streams with no host synchronisation."""
    import torch

    from protocols.distributed_keygen_amd import limbs as L

    rng = random.Random(777)
    n = rng.getrandbits(1027) | (1 << 1026) | 1
    n2 = n * n
    exps = [rng.getrandbits(400) | 1 for _ in range(3)]
    bases = [rng.randrange(n2) for _ in range(96)]
    rows = eng.to_device(L.pack(bases, L.limbs_for(n2)))
    streams = [torch.cuda.Stream() for _ in range(4)]
    torch.cuda.synchronize()
    outs = []
    for k in range(8):
        with torch.cuda.stream(streams[k % 4]):
            e = exps[k % 3]
            out = eng.powmod_nsquare_t(rows, n, e) if k % 2 == 0 else eng.powmod_shared_t(rows, n2, e)
            outs.append((e, out))
    torch.cuda.synchronize()
    assert len(eng._ws) >= 4
    for e, out in outs:
        assert L.unpack(eng.to_host(out)) == [pow(b, e, n2) for b in bases]


@pytest.mark.parametrize("segments", [1, 2, 3, 4, 7, 64])
def test_powmod_nsquare_segments_are_bit_identical(eng, segments):
    """One exponentiation enqueued as several consecutive launches (tape segments, the accumulator
    travelling through the workspace): the same result bit for bit, in every launch shape of the pair kernel."""
    rng = random.Random(1000 + segments)
    n = rng.getrandbits(1027) | (1 << 1026) | 1
    n2 = n * n
    bases = [0, 1, n, n2 - 1] + [rng.randrange(n2) for _ in range(29)]
    try:
        for lpl, wpg in ((9, 1), (18, 1), (3, 2), (9, 2), (18, 2)):
            eng.set_limbs_per_lane(lpl)
            eng.set_wavefronts_per_group(wpg)
            for e in (rng.getrandbits(700) | (1 << 699) | 1, (1 << 300), 5, 0):
                eng.set_segments(segments)
                assert eng.powmod_nsquare_batch(bases, e, n) == [pow(b, e, n2) for b in bases], (lpl, wpg, e.bit_length())
        # the developer knob overrides the library's automatic choice (segments = 0) and nothing else
        eng.set_segments(0)
        eng.debug_knob("n2_segments", segments)
        e = rng.getrandbits(600) | 1
        assert eng.powmod_nsquare_batch(bases, e, n) == [pow(b, e, n2) for b in bases]
    finally:
        eng.debug_knob("n2_segments", 0)
        eng.set_segments(0)
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)


@pytest.mark.parametrize("n_bits,batch,exp_bits", [(131, 70000, 300), (131, 40, 300), (515, 20000, 200), (1027, 33, 700),
                                                   (2051, 5200, 70), (2051, 9, 2100), (4099, 1500, 40), (4099, 5, 600)])
def test_powmod_nsquare_timesliced_is_bit_identical(eng, n_bits, batch, exp_bits):
    """The time-sliced form of the two-wavefront kernel (resident workgroups that take (segment, group) units from a
    ticket counter; csrc/mx_powmod_n2_split.hpp): forced on with the developer knob, with more groups than resident
    pairs (the large batches) and fewer, for every segment count — bit for bit the plain launch's result."""
    rng = random.Random(n_bits * 31 + batch)
    n = rng.getrandbits(n_bits) | (1 << (n_bits - 1)) | 1
    n2 = n * n
    bases = [0, 1, n, n + 1, n2 - 1][: batch] + [rng.randrange(n2) for _ in range(max(0, batch - 5))]
    e = rng.getrandbits(exp_bits) | (1 << (exp_bits - 1)) | 1
    want = [pow(b, e, n2) for b in bases]
    try:
        eng.debug_knob("n2_timeslice", 2)
        for lpl in (9, 18):
            eng.set_limbs_per_lane(lpl)
            eng.set_wavefronts_per_group(2)
            for segments in ((0, 1, 3) if batch < 1000 else (0, 2)):
                eng.set_segments(segments)
                assert eng.powmod_nsquare_batch(bases, e, n) == want, (lpl, segments)
        eng.debug_knob("n2_timeslice", 1)
        eng.set_segments(0)
        assert eng.powmod_nsquare_batch(bases, e, n) == want
    finally:
        eng.debug_knob("n2_timeslice", 0)
        eng.set_segments(0)
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)


def test_exponent_with_a_very_long_run_of_zero_bits(eng):
    """the squaring count of a schedule step used to be packed into 16 bits, so an
    exponent with >= 65536 consecutive zero bits (2^70000) silently gave a wrong power.  Both the
    generic sliding-window kernel and the N^2 tape must handle it."""
    rng = random.Random(70000)
    mod = rng.getrandbits(120) | (1 << 119) | 1
    n = rng.getrandbits(60) | (1 << 59) | 1
    bases = [2, 3, mod - 1] + [rng.randrange(mod) for _ in range(5)]
    for e in (1 << 70000, (1 << 70000) + 1, (1 << 131072) + (1 << 3)):
        assert eng.powmod_batch(bases, e, mod) == [pow(b, e, mod) for b in bases], e.bit_length()
        nb = [b % (n * n) for b in bases]
        assert eng.powmod_nsquare_batch(nb, e, n) == [pow(b, e, n * n) for b in nb], e.bit_length()


def test_powmod_nsquare_split_launch_is_bit_identical(eng):
    """One batch above the capacity of the wide two-wavefront shape runs as two launches side by side on the engine's
    companion stream (mx_nsquare_launch_split; since round 5 only when the developer knob asks — the time-sliced wide
    launch is as fast): same rows as the single launch and as pow(), also on a ragged remainder; explicit shapes never split."""
    import torch

    from protocols.distributed_keygen_amd import limbs as L, synthetic

    key = synthetic.make_key(2048, 3, 1)
    n, n2 = key.n, key.n_square
    e = (1 << 70) + 12345
    eng.set_limbs_per_lane(0)
    eng.set_wavefronts_per_group(0)
    try:
        assert eng.nsquare_launch_split(n.bit_length(), 8192) is None and eng.nsquare_launch_split(n.bit_length(), 10000) is None
        assert eng.nsquare_launch_split(n.bit_length(), 11264) is None and eng.nsquare_launch_split(n.bit_length(), 12288) is None
        eng.debug_knob("n2_split", 2)
        assert eng.nsquare_launch_split(n.bit_length(), 11264) == (8192, (18, 2), (9, 2))
        for batch, knob in ((11264, 2), (8192 + 333, 2)):
            cts = synthetic.random_ciphertexts(key, batch, seed=batch)
            rows = eng.to_device(L.pack(cts, L.limbs_for(n2)))
            eng.debug_knob("n2_split", knob)
            assert eng.nsquare_launch_split(n.bit_length(), batch) is not None
            got = eng.powmod_nsquare_t(rows, n, e)
            eng.debug_knob("n2_split", 1)
            assert eng.nsquare_launch_split(n.bit_length(), batch) is None
            single = eng.powmod_nsquare_t(rows, n, e)
            torch.cuda.synchronize()
            assert torch.equal(got, single)
            idx = [0, 1, 8191, 8192, 8193, batch - 1]
            assert L.unpack(eng.to_host(got[idx])) == [pow(cts[k], e, n2) for k in idx]
        eng.debug_knob("n2_split", 2)
        eng.set_limbs_per_lane(9)
        assert eng.nsquare_launch_split(n.bit_length(), 11264) is None          # an explicit shape is the caller's choice
    finally:
        eng.debug_knob("n2_split", 0)
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)


def test_fixed_window_tape_is_bit_identical_and_secret_independent_in_shape(eng):
    """MX_PLAN_FIXED_WINDOW: the tape of a partial decryption with fixed windows — one multiplication
    per window whatever the digits, zero windows multiplying by the domain's one — gives the same results as the
    sliding-window tape in every launch shape and with segments, and its operation counts are a function of the
    exponent's LENGTH only (two exponents of one length with very different bits: identical counts; the sliding tape
    differs between them)."""
    rng = random.Random(4197)
    n = rng.getrandbits(1027) | (1 << 1026) | 1
    n2 = n * n
    bases = [0, 1, n, n + 1, n2 - 1] + [rng.randrange(n2) for _ in range(28)]
    e_dense = (1 << 900) - 1                                  # all ones
    e_sparse = (1 << 899) | 1                                 # two ones, 898 zero bits between them
    e_random = rng.getrandbits(900) | (1 << 899)
    e_zero_windows = ((rng.getrandbits(300) | (1 << 299)) << 600) | rng.getrandbits(100)      # a long run of zero windows
    try:
        eng.set_fixed_window(True)
        shapes = {e: (eng.nsquare_plan(n, e).desc.n_sqr, eng.nsquare_plan(n, e).desc.n_mul, eng.nsquare_plan(n, e).desc.ntape)
                  for e in (e_dense, e_sparse, e_random, e_zero_windows)}
        assert len(set(shapes.values())) == 1, shapes           # schedule = f(bit length) only
        for lpl, wpg, seg in ((0, 0, 0), (9, 1, 1), (18, 1, 4), (3, 2, 1), (9, 2, 3), (18, 2, 1)):
            eng.set_limbs_per_lane(lpl)
            eng.set_wavefronts_per_group(wpg)
            eng.set_segments(seg)
            for e in (e_dense, e_sparse, e_random, e_zero_windows, 5, 1, 0, (1 << 64) + 1):
                assert eng.powmod_nsquare_batch(bases, e, n) == [pow(b, e, n2) for b in bases], (lpl, wpg, seg, e.bit_length())
        eng.set_fixed_window(False)
        sliding = {e: eng.nsquare_plan(n, e).desc.n_mul for e in (e_dense, e_sparse)}
        assert sliding[e_dense] != sliding[e_sparse]            # the default tape does depend on the bits
        fixed_cost = shapes[e_dense][1]
        assert sliding[e_sparse] < fixed_cost                   # and is never dearer
    finally:
        eng.set_fixed_window(False)
        eng.set_segments(0)
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)
    # key_length 2048 at the headline's exponent length: the cost the header states (728 vs ~592 multiplications)
    key_n = rng.getrandbits(2051) | (1 << 2050) | 1
    e = rng.getrandbits(4197) | (1 << 4196)
    cts = [rng.randrange(key_n * key_n) for _ in range(9)]
    try:
        eng.set_fixed_window(True)
        d = eng.nsquare_plan(key_n, e).desc
        assert d.window == 8 and 720 <= d.n_mul <= 735 and d.n_sqr <= 4197     # w = 7: 126 table products + 599 windows + 3
        assert eng.powmod_nsquare_batch(cts, e, key_n) == [pow(c, e, key_n * key_n) for c in cts]
    finally:
        eng.set_fixed_window(False)


@pytest.mark.parametrize("bits", [50, 131, 300, 600, 1029, 2050, 2053, 4100, 5359])
def test_bipartite_latency_form(eng, bits):
    """The bipartite form of the generic kernel (limbs_per_lane 6, csrc/mx_bimont.hpp: every product split over two
    wavefronts, least-significant-first Montgomery steps on one, most-significant-first steps with a fold on the other):
    bit-exact against pow() on special and random moduli, bases and exponents, per-group and shared exponents, ragged
    batches, and the biprimality-test shape (40 bases per candidate)."""
    rng = random.Random(bits * 31)
    eng.set_limbs_per_lane(6)
    try:
        assert eng.generic_launch_form(bits, 80, 2)[0] == 2
        specials = [(1 << bits) - 1, (1 << (bits - 1)) + 1, ((1 << bits) - 1) ^ (1 << (bits // 2)), (1 << bits) - (1 << (bits // 3)) - 1]
        mods = [m | 1 for m in specials] + [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(3)]
        exps = [(1 << min(bits, 70)) - 1, 1 << min(bits - 1, 60), 0, 1, 2, rng.getrandbits(min(bits, 150)) | 1, rng.getrandbits(bits)]
        pat29 = lambda m: sum(((1 << 29) - 1) << (29 * k) for k in range(0, bits // 29 + 1, 2)) % m
        rows = [[0, 1, m - 1, m // 2, pat29(m), (m - pat29(m)) % m] + [rng.randrange(m) for _ in range(5)] for m in mods]
        assert eng.powmod_batch_multi(rows, exps, mods) == [[pow(b, e, m) for b in r] for r, e, m in zip(rows, exps, mods)]
        # one exponent for the whole launch (mx_powmod_shared_lpl), a ragged batch
        m, e = mods[-1], rng.getrandbits(min(bits, 200)) | 1
        bases = [rng.randrange(m) for _ in range(13)] + [0, 1, m - 1]
        assert eng.powmod_batch(bases, e, m) == [pow(b, e, m) for b in bases]
        if bits in (1029, 2053):
            # the shape it exists for: a keygen round's survivors, 40 bases each, full-length and half-length exponents
            cands = [rng.getrandbits(bits) | (1 << (bits - 1)) | 1 for _ in range(7)]
            gens = [[rng.randrange(c) for _ in range(40)] for c in cands]
            for ebits in (bits - 2, bits // 2 + 3):
                ex = [rng.getrandbits(ebits) | (1 << (ebits - 1)) for _ in cands]
                assert eng.powmod_batch_multi(gens, ex, cands) == [[pow(g, e, c) for g in gs] for gs, e, c in zip(gens, ex, cands)]
    finally:
        eng.set_limbs_per_lane(0)


def test_bipartite_form_with_moduli_of_different_lengths_in_one_launch(eng):
    """A launch has ONE geometry (from its longest modulus) and every group its own modulus: the factor that leaves the
    domain must be reduced modulo EACH group's N.  The first form used the power of two 2^(W(Pd - pivot)) itself, which the
    geometry keeps below a modulus of the launch's bit length only; for (m - 1)^even modulo a modulus five bits shorter
    the epilogue then returned m + 1 (tools/soak_round5.py seed 19, reproduced below with the pivot it had), and with the
    library's own pivot any modulus below about half the launch's length broke the same way."""
    rng = random.Random(1905)
    eng.set_limbs_per_lane(6)
    try:
        mods = [0x71fbad03ca2e35fc42df03f, 0x3ffffffffffffffbfffffff, 0x158572c1f2c53615402469d, 0x9a8a626a55a009bae63c6f,
                0x7fb7e00a02f6bbf3437e4f, 0x2feaaa2629da259a2420d1, 0x1ffffffffffbffffffffff]
        exps = [0xa4bc2ca7c8e97fdd, 1, 0xffffffffffffffff, 0, 1, 0xee202d3b477e5540, 0xffffffffffffffff]
        rows = [[0], [1], [0xff926be0a485352718e705], [1], [1], [0x2feaaa2629da259a2420d0], [0x1ffffffffffbfffffffffe]]
        for pivot in (3, 0):
            eng.debug_knob("bi_pivot", pivot)
            assert eng.powmod_batch_multi(rows, exps, mods) == [[pow(b, e, m) for b in r] for r, e, m in zip(rows, exps, mods)], pivot
        # moduli from full length down to a few limbs in one launch, at the key lengths the form is chosen for, every pivot class
        for bits in (1029, 2053):
            lens = [bits, bits - 1, bits - 7, bits - 40, bits * 3 // 4, bits // 2 + 1, bits // 2 - 30, bits // 3, 200, 61, 3]
            mods = [rng.getrandbits(b) | (1 << (b - 1)) | 1 for b in lens]
            exps = [rng.getrandbits(rng.choice([bits - 2, 64, 300])) | 1 for _ in mods]
            exps[0] &= ~1
            rows = [[m - 1, 1, 0, rng.randrange(m), rng.randrange(m)] for m in mods]
            want = [[pow(b, e, m) for b in r] for r, e, m in zip(rows, exps, mods)]
            steps = 3 * (-(-(bits + 35) // 87)) + 3
            for pivot in (0, 3, 3 * (steps // 6), steps - 6):
                eng.debug_knob("bi_pivot", pivot)
                assert eng.generic_launch_form(bits, 5 * len(mods), len(mods))[0] == 2
                assert eng.powmod_batch_multi(rows, exps, mods) == want, (bits, pivot)
            eng.debug_knob("bi_pivot", 0)
            for lpl in (3, 9, 18, 0):                 # the same launch in the one-wavefront forms (and the library's choice)
                eng.set_limbs_per_lane(lpl)
                assert eng.powmod_batch_multi(rows, exps, mods) == want, (bits, lpl)
            eng.set_limbs_per_lane(6)
    finally:
        eng.debug_knob("bi_pivot", 0)
        eng.set_limbs_per_lane(0)


@pytest.mark.parametrize("n_bits", [1027, 1029, 2050, 2051, 2053, 1500, 900, 4099, 3100])
def test_four_wavefront_latency_form_of_the_pair_kernel(eng, n_bits):
    """csrc/mx_bipair.hpp (round 6): both passes of every pair product bipartite, four wavefronts per group of elements and a fifth for the quotient correction —
    what a lone decrypt() and every launch of at most one workgroup per compute unit run at key_length 1024 / 2048.
    Random and special moduli, bases that are 0 / 1 / multiples of N / N^2 - 1, exponents of 1 bit .. full length (incl. the
    fixed-window tape), batches from 1 to more than one workgroup per CU: bit for bit CPython pow; the library's own choice
    takes the form for small batches and the developer knob switches that off."""
    rng = random.Random(n_bits * 7 + 1)
    try:
        for trial, n in enumerate([rng.getrandbits(n_bits) | (1 << (n_bits - 1)) | 1, (1 << n_bits) - 1, (1 << (n_bits - 1)) + 1]):
            n2 = n * n
            for ebits, batch in ((1, 3), (2, 1), (3, 5), (64, 2), (300, 70), (n_bits, 9), (2 * n_bits + 90, 4)):
                e = rng.getrandbits(ebits) | (1 << (ebits - 1))
                bases = ([0, 1, n, n2 - 1, n + 1, n * (n - 1)] + [rng.randrange(n2) for _ in range(batch)])[:max(batch, 1)]
                want = [pow(b, e, n2) for b in bases]
                eng.set_limbs_per_lane(3)
                eng.set_wavefronts_per_group(4)
                assert eng.nsquare_launch_shape(n_bits, len(bases))[4] == 4
                assert eng.powmod_nsquare_batch(bases, e, n) == want, (trial, ebits, batch)
                if trial == 0 and ebits in (64, n_bits):
                    eng.set_fixed_window(True)                       # the fixed-window tape through the same kernel
                    assert eng.powmod_nsquare_batch(bases, e, n) == want
                    eng.set_fixed_window(False)
                    eng.set_limbs_per_lane(0)
                    eng.set_wavefronts_per_group(0)                  # the library's choice for a launch this small: this form
                    assert eng.nsquare_launch_shape(n_bits, len(bases))[4] == 4
                    assert eng.powmod_nsquare_batch(bases, e, n) == want
                    eng.debug_knob("n2_bipair", 1)
                    assert eng.nsquare_launch_shape(n_bits, len(bases))[4] == 2
                    assert eng.powmod_nsquare_batch(bases, e, n) == want
                    eng.debug_knob("n2_bipair", 0)
        # more than one workgroup per compute unit (the form is then slower than two wavefronts, but must stay right)
        n = rng.getrandbits(n_bits) | (1 << (n_bits - 1)) | 1
        bases = [rng.randrange(n * n) for _ in range(1500)]
        e = rng.getrandbits(40) | 1
        eng.set_limbs_per_lane(3)
        eng.set_wavefronts_per_group(4)
        assert eng.powmod_nsquare_batch(bases, e, n) == [pow(b, e, n * n) for b in bases]
    finally:
        eng.set_fixed_window(False)
        eng.debug_knob("n2_bipair", 0)
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)


def test_four_wavefront_form_exists_only_where_its_kernel_does(eng):
    """Groups of 16, 32 and 64 lanes (moduli of ~800 .. 2560 and ~2800 .. 5500 bits: key_length 1024, 2048, 4096); an explicit
    request elsewhere — below, and between the ranges, where the two geometries the form combines disagree — is refused, the
    library's own choice falls back to two wavefronts."""
    from protocols.distributed_keygen_amd._lib import MxError

    rng = random.Random(3)
    for n_bits in (300, 2600):
        n = rng.getrandbits(n_bits) | (1 << (n_bits - 1)) | 1
        bases = [rng.randrange(n * n) for _ in range(3)]
        try:
            eng.set_limbs_per_lane(3)
            eng.set_wavefronts_per_group(4)
            with pytest.raises(MxError):
                eng.powmod_nsquare_batch(bases, 65537, n)
            eng.set_limbs_per_lane(0)
            eng.set_wavefronts_per_group(0)
            assert eng.nsquare_launch_shape(n_bits, 3)[4] == 2
            assert eng.powmod_nsquare_batch(bases, 65537, n) == [pow(b, 65537, n * n) for b in bases]
        finally:
            eng.set_limbs_per_lane(0)
            eng.set_wavefronts_per_group(0)
