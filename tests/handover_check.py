"""Hot hand-overs of a time-sliced launch (csrc/mx_powmod_n2_split.hpp, the unit queues) — the checker shared by
tests/test_gpu_handover.py (inside `pytest -m gpu`) and tools/ts_handover_check.py (by hand, any configuration).

The configuration that exposes a broken release: as many RESIDENT wavefront pairs as groups (or more), so that a free pair
is already polling when a group's next unit is published and reads what the previous pair stored the moment the entry
appears.  Rounds 3-4 shipped a publish sequence that did not wait for the L2 write-back (the compiler's release store
behind an atomic whose result had been waited for); one group in ~1000 hand-overs came out wrong, and no test with more
groups than resident pairs (where a group is mostly handed back to the pair that pushed it) ever saw it.

Not a test module (no test_ prefix): imported only."""

from __future__ import annotations

from typing import Callable, List, Optional, Tuple

SHAPES = {
    # (batch, limbs per lane, resident workgroups per CU, units per group): profiles/r05_ts_handover_check.txt — on the
    # -DMX_DEV_TS_COMPILER_RELEASE build the first loses 24-72 rows in EVERY run, the second 8 rows in some runs
    "k2048_l9_r3_u8": (2048, 10000, 9, 3, 8),
    "k2048_l9_r2_u4": (2048, 8192, 9, 2, 4),
    "k2048_l18_r1_u8": (2048, 8192, 18, 1, 8),
    "k2048_l18_r1_u16": (2048, 6000, 18, 1, 16),
    "k4096_l9_r2_u8": (4096, 4000, 9, 2, 8),
    "k4096_l18_r1_u8": (4096, 4096, 18, 1, 8),
}

_inputs = {}


def _inputs_for(eng, key_length: int, batch: int):
    """(n, exp, device rows of `batch` ciphertexts, rows of the plain one-wavefront launch) — cached per key length."""
    import torch

    from protocols.distributed_keygen_amd import limbs as L, synthetic

    hit = _inputs.get(key_length)
    if hit is None or hit[2].shape[0] < batch:
        key = synthetic.make_key(key_length, 3, 1)
        own = next(i for i in (1, 2, 3) if key.exponent(i) > 0)
        exp, n = key.exponent(own), key.n
        rows = max(batch, 10000 if key_length <= 2048 else 4096)
        c = eng.to_device(L.pack(synthetic.random_ciphertexts(key, rows, seed=7), L.limbs_for(key.n_square)))
        eng.set_limbs_per_lane(18)
        eng.set_wavefronts_per_group(1)
        eng.debug_knob("n2_timeslice", 1)            # 1: never time-sliced
        want = eng.powmod_nsquare_t(c, n, exp, segments=1).clone()
        torch.cuda.synchronize()
        hit = _inputs[key_length] = (n, exp, c, want)
    n, exp, c, want = hit
    return n, exp, c[:batch], want[:batch]


def queue_words_consistent(sched, groups: int, units: int) -> bool:
    """The scheduling words a time-sliced launch leaves behind: level 0 granted to at least every group (its
    fetch-and-add head may overshoot), every other level granted AND reserved exactly `groups` times, and each level's
    ring a permutation of the groups — every unit handed out once and written once."""
    heads, tails = list(sched[:units]), list(sched[16:16 + units])
    ok = heads[0] >= groups and heads[1:] == [groups] * (units - 1) and tails[1:] == [groups] * (units - 1)
    for lv in range(1, units):
        ring = sched[32 + (lv - 1) * groups: 32 + lv * groups]
        ok = ok and sorted(int(x) for x in ring) == list(range(1, groups + 1))
    return bool(ok)


def check(eng, key_length: int, batch: int, lpl: int, resident: int, units: int, reps: int = 10,
          log: Optional[Callable[[str], None]] = None) -> Tuple[int, int, List[int]]:
    """Runs the time-sliced launch `reps` times in the hot configuration; returns (wrong rows in total, runs whose queue
    words were inconsistent, wrong rows per run).  The engine's launch-shape settings are restored."""
    import torch

    n, exp, c, want = _inputs_for(eng, key_length, batch)
    try:
        eng.set_limbs_per_lane(lpl)
        eng.set_wavefronts_per_group(2)
        eng.debug_knob("n2_timeslice", 16 + resident)          # time-sliced with `resident` workgroups per CU, whatever the size
        assert eng.nsquare_launch_timesliced(n.bit_length(), batch)[0] == resident, "this shape has no time-sliced instance"
        k, l, _w, _blocks, _wf = eng.nsquare_launch_shape(n.bit_length(), batch)
        gpw = 64 // k
        groups, nblocks = (batch + gpw - 1) // gpw, (batch + 2 * gpw - 1) // (2 * gpw)
        pairs = 2 * resident * torch.cuda.get_device_properties(eng.device).multi_processor_count
        nslots = 8 + (1 << (eng.nsquare_plan(n, exp).desc.window - 1))               # csrc/mx_capi_n2.hip: shape_n2
        table_bytes = (nslots * 2 * l * nblocks * 128 * 4 + 255) // 256 * 256       # the queues sit behind the pair table
        wrong, per_run, bad_queues = 0, [], 0
        for it in range(reps):
            out = eng.powmod_nsquare_t(c, n, exp, segments=units)
            torch.cuda.synchronize()
            bad = (out != want).any(dim=1).nonzero().flatten().cpu().numpy()
            ws = eng._ws[eng._stream_ptr()]
            sched = ws[table_bytes: table_bytes + (32 + groups * (units - 1)) * 4].view(torch.int32).cpu().numpy()
            ok = queue_words_consistent(sched, groups, units)
            wrong += len(bad)
            per_run.append(len(bad))
            bad_queues += 0 if ok else 1
            if log:
                log(f"run {it}: {len(bad)} wrong rows (groups {sorted(set(int(x) // gpw for x in bad))[:12]}); queues consistent: {ok}")
        if log:
            log(f"key_length {key_length}, batch {batch}, L{l} K={k}: {groups} groups on {pairs} resident pairs ({resident} workgroup(s) per CU), "
                f"{units} units per group, {reps} runs: {wrong} wrong rows, {bad_queues} runs with inconsistent queues")
        return wrong, bad_queues, per_run
    finally:
        eng.debug_knob("n2_timeslice", 0)
        eng.set_limbs_per_lane(0)
        eng.set_wavefronts_per_group(0)
