"""Batched GPU operators over the C ABI (include/mxpaillier.h).

Two levels:
  * tensor level (``*_t``): operands are device-resident ``torch.int32`` tensors of limb rows
    ``[batch, limbs]`` — what bench.py and the multi-GPU path use;
  * int level: Python ints in / out, mirroring the reference's scalar operators
    (``pow_mod(value, exponent, modulus)`` of tno.mpc.encryption_schemes.utils, bound at
    distributed_keygen.py:35 and paillier_shared_key.py:20) as the batched forms
    ``powmod_batch`` / ``powmod_batch_multi`` / ``sieve_batch`` / ``combine_batch`` /
    ``biprime_verdict_batch``.

PyTorch is used only for device memory and streams.  There is no CPU fallback: constructing an
Engine without a GPU or without the built library raises.
"""

from __future__ import annotations

from collections import OrderedDict
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib, limbs as _limbs


def _int_args(fn):
    """Coerce the integer operands of an entry point once: parameters annotated ``int`` become Python ints and the
    per-group operand lists (``mods``, ``exps``, ``coeffs``, ``primes`` annotated ``Sequence[int]``) lists of Python
    ints, so that int-like values — gmpy2.mpz when the reference's utils run on gmpy2, numpy integers — behave like the
    ints the reference passes (its own leaf accepts them)."""
    import functools
    import inspect

    params = inspect.signature(fn).parameters
    names = list(params)
    scalars = {k for k, q in params.items() if q.annotation == "int"}
    lists = {k for k, q in params.items() if q.annotation == "Sequence[int]" and k in ("mods", "exps", "coeffs", "primes")}
    if not scalars and not lists:
        return fn

    def conv(name, v):
        if name in scalars:
            return v if type(v) is int else int(v)
        if name in lists and not (isinstance(v, list) and all(type(x) is int for x in v[:1])):
            return [int(x) for x in v]
        return v

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        args = tuple(conv(names[i], a) if i < len(names) else a for i, a in enumerate(args))
        kwargs = {k: conv(k, v) for k, v in kwargs.items()}
        return fn(*args, **kwargs)

    return wrapper


class _Plan:
    """A per-key plan: the ctypes descriptor, the device block it points to (kept alive here) and the
    event that marks the end of prepare's uploads on the stream they were enqueued on."""

    __slots__ = ("desc", "block", "stream_ptr", "ready")

    def __init__(self, desc, block, stream_ptr, ready) -> None:
        self.desc, self.block, self.stream_ptr, self.ready = desc, block, stream_ptr, ready


class _ModulusRows:
    """Moduli of a key-generation round kept on the device between its steps (survivors of the sieve, in order):
    device rows of the width they were produced in, and the largest bit length."""

    __slots__ = ("rows", "bits")

    def __init__(self, rows, bits: int) -> None:
        self.rows, self.bits = rows, int(bits)

    def operand(self, eng, groups: int, limbs: int):
        """The (rows, bits) operand the tensor-level entry points take, `limbs` words wide."""
        if self.rows.shape[0] != groups:
            raise ValueError("the kept moduli rows do not belong to these candidates")
        if self.bits > 32 * limbs:
            raise ValueError("modulus wider than the limb rows")
        rows = self.rows
        if rows.shape[1] > limbs:
            rows = rows[:, :limbs].contiguous()         # produced in the Shamir field's width: the upper words are zero
        elif rows.shape[1] < limbs:
            rows = eng.torch.nn.functional.pad(rows, (0, limbs - rows.shape[1]))
        return rows, self.bits


    def repeated(self, times: int) -> "_ModulusRows":
        """The same moduli `times` times over (the candidate groups of co-located parties in one launch)."""
        return self if times == 1 else _ModulusRows(self.rows.repeat(times, 1), self.bits)


class _VRows:
    """A party's v values of a round kept on the device: rows [groups * keep, limbs] as biprime_v_t produced them
    (rows beyond a candidate's count hold the modexp of a zero row: 0 or 1) and the counts."""

    __slots__ = ("rows", "counts", "keep")

    def __init__(self, rows, counts, keep: int) -> None:
        self.rows, self.counts, self.keep = rows, counts, int(keep)

    def slots(self, eng, groups: int, n_slots: int, limbs: int):
        if self.rows.shape[0] != groups * self.keep or n_slots > self.keep or self.rows.shape[1] != limbs:
            raise ValueError("the kept v rows do not belong to these candidates")
        return self.rows.view(groups, self.keep, limbs)[:, :n_slots, :]

    def part(self, k: int, parts: int) -> "_VRows":
        """The k-th of `parts` equal slices of the candidate groups (one party's share of a merged launch)."""
        groups = len(self.counts) // parts
        return _VRows(self.rows[k * groups * self.keep : (k + 1) * groups * self.keep], self.counts[k * groups : (k + 1) * groups], self.keep)


class NestedColumn:
    """A party's v values for ``Engine.biprime_verdict_columns`` as ONE LIST PER CANDIDATE (what the reference exchanges,
    distributed_keygen.py:1331-1337) instead of one flat list: each candidate contributes its first n_slots values, zero
    padded — packed by one codec call without a flattened copy."""

    __slots__ = ("lists",)

    def __init__(self, lists) -> None:
        self.lists = lists if isinstance(lists, (list, tuple)) else list(lists)


class Engine:
    """One engine per process/GPU.  Calls enqueue on ``torch.cuda.current_stream()``; every stream gets
    its own workspace, so one Engine may be driven from several streams (one launch in flight per
    stream).  Not thread-safe (matches the reference's single asyncio thread)."""

    MAX_PLANS = 16     # per-key plans kept (a party normally has one key)
    NestedColumn = NestedColumn      # biprime.py asks the engine it was given for it (the CPU test double has none)

    def __init__(self, device: Optional[int] = None) -> None:
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("protocols.distributed_keygen_amd needs an AMD GPU (torch.cuda unavailable); no CPU fallback")
        self.torch = torch
        self.lib = _lib.lib()
        idx = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", idx)
        self._ws: Dict[int, Any] = {}          # stream -> workspace tensor
        self._lpl = 0                          # lane geometry of this engine's modexp launches (0 = automatic)
        self._wpg = 0                          # wavefronts per group of the N^2 pair kernel (0 = automatic, 1, 2)
        self._segments = 0                     # launches per N^2 exponentiation (0 = automatic; include/mxpaillier.h)
        self._fixed_window = False             # tapes of new N^2 plans: fixed windows (secret-independent schedule) instead of sliding ones
        self._n2_plans: "OrderedDict[Tuple[int, int, bool], _Plan]" = OrderedDict()
        self._combine_plans: "OrderedDict[Tuple[int, int, int], _Plan]" = OrderedDict()
        self._side_streams: List[Any] = []     # chunked int-level batches (_pipelined): streams verified concurrent
        self._side_streams_capped = False      # the process has fewer concurrent queues than chunks were wanted
        self._pin: Dict[str, Any] = {}         # pinned staging buffers of _pipelined
        self.last_timing: Optional[Dict[str, Any]] = None   # host/GPU time split of the last int-level modexp batch
        self._priority_aux = False             # small kernels on a high-priority companion stream (set_priority_aux)
        self._aux: Dict[int, Any] = {}         # stream -> its companion
        self._split_streams: Dict[int, Any] = {}   # stream -> the companion that runs the second part of a split launch

    # ------------------------------------------------------------------ plumbing
    def _stream_ptr(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _workspace(self, nbytes: int):
        """Scratch of the CURRENT stream (launches on different streams must not share scratch: the
        window tables of one launch would be overwritten by the next)."""
        if nbytes < 0:
            _lib.check(int(nbytes), "workspace query")
        key = self._stream_ptr()
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            self._ws[key] = None
            with self.torch.cuda.device(self.device):
                ws = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def _use_plan(self, plan: _Plan) -> None:
        """Order the current stream after the plan's uploads if they were enqueued on another stream."""
        cur = self.torch.cuda.current_stream(self.device)
        if plan.stream_ptr != int(cur.cuda_stream):
            # the block was allocated on the stream that prepared it: tell the caching allocator that this
            # stream reads it too, so that an evicted plan's memory is not handed out while launches of
            # this stream that use it are still in flight
            plan.block.record_stream(cur)
        if plan.ready is not None:
            if plan.stream_ptr != int(cur.cuda_stream):
                cur.wait_event(plan.ready)
            if plan.ready.query():
                plan.ready = None

    def _cache_plan(self, cache: "OrderedDict[Any, _Plan]", key: Any, plan: _Plan) -> _Plan:
        """Keeps `plan`, dropping the least recently used ones beyond MAX_PLANS.  Dropping a plan only drops this
        engine's reference to its device block; the memory goes back to torch's caching allocator, which hands a block
        out again only behind the work of (a) the stream it was allocated on — the stream that prepared the plan — and
        (b) every stream named to it with record_stream.  _use_plan names every OTHER stream before that stream's launch
        reads the plan, so a launch still in flight on any stream when a 17th key evicts its plan keeps reading intact
        memory (tests/test_host_logic.py: the eviction test on a stand-in allocator)."""
        cache[key] = plan
        while len(cache) > self.MAX_PLANS:
            cache.popitem(last=False)
        return plan

    def to_device(self, rows: np.ndarray):
        """uint32 rows -> int32 device tensor (bit pattern preserved)."""
        t = self.torch.from_numpy(np.ascontiguousarray(rows, dtype="<u4").view(np.int32))
        return t.to(self.device, non_blocking=False)

    @staticmethod
    def to_host(t) -> np.ndarray:
        return t.detach().cpu().numpy().view(np.uint32)

    def synchronize(self) -> None:
        self.torch.cuda.current_stream(self.device).synchronize()

    def set_limbs_per_lane(self, limbs_per_lane: int) -> None:
        """Lane geometry of this engine's modexp launches: 0 = automatic (from the batch size), 9 =
        narrow, 18 = wide, 3 = the latency geometry (the N^2 pair kernel: two wavefronts per group, any key
        length; the generic-modulus kernels: moduli up to 5533 bits, wider ones fall back to the automatic
        choice), 6 = the generic kernels' bipartite latency form (3 limbs per lane, every product on two wavefronts:
        moduli up to 5359 bits; the N^2 pair kernel reads it as 3).  Passed with every call (no process-wide state)."""
        if limbs_per_lane not in (0, 3, 6, 9, 18):
            raise ValueError("limbs_per_lane must be 0, 3, 6, 9 or 18")
        self._lpl = int(limbs_per_lane)

    def set_wavefronts_per_group(self, wavefronts: int) -> None:
        """N^2 pair kernel (powmod_nsquare_t): 1 = one wavefront runs both Montgomery passes of a pair product,
        2 = two wavefronts, one pass each (for launches that leave SIMDs idle), 4 = the five-wavefront latency form
        (both passes bipartite, csrc/mx_bipair.hpp: 3 limbs per lane only, moduli whose groups have 16 or 32 lanes),
        0 = the library's choice (include/mxpaillier.h: mx_powmod_nsquare_run)."""
        if wavefronts not in (0, 1, 2, 4):
            raise ValueError("wavefronts per group must be 0, 1, 2 or 4")
        self._wpg = int(wavefronts)

    GENERIC_LATENCY_MAX_BITS = 29 * 3 * 64 - 4 - 31      # 5533: widest modulus with a 3-limb generic instance (mx_host.hpp: choose_geometry)

    def _lpl_generic(self, mod_bits: int = 0) -> int:
        """The engine's lane geometry as the generic-modulus kernels take it.  The latency geometry (3) exists for the
        N^2 pair kernel at every key length but for a GENERIC modulus only up to 5533 bits: an engine tuned with
        set_limbs_per_lane(3) for low-latency decryptions leaves wider generic launches (N^2 of key_length 4096 through
        powmod_batch, say) to the library's automatic choice instead of failing with MX_ERR_SIZE."""
        if self._lpl == 3 and mod_bits > self.GENERIC_LATENCY_MAX_BITS:
            return 0
        if self._lpl == 6 and mod_bits > self.GENERIC_BIPARTITE_MAX_BITS:
            return 0
        return self._lpl if self._lpl in (3, 6, 9, 18) else 0

    GENERIC_BIPARTITE_MAX_BITS = 29 * 3 * 62 - 35        # 5359: wavefront H needs Pd / 3 + 2 <= 64 lanes (mx_host.hpp)

    def _lpl_n2(self) -> int:
        """... as the N^2 pair kernel takes it: 6 (the generic kernel's bipartite latency form) is its latency geometry 3."""
        return 3 if self._lpl == 6 else self._lpl

    def generic_launch_form(self, mod_bits: int, batch: int = 1, groups: int = 1) -> Tuple[int, int]:
        """(wavefronts per group of elements, pivot) of a generic-modulus modexp launch with this engine's settings:
        (2, hL) for the bipartite latency form (include/mxpaillier.h: mx_powmod_launch_form), else (1, 0)."""
        import ctypes

        waves, pivot = ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.mx_powmod_launch_form(mod_bits, batch, groups, self._lpl_generic(mod_bits), waves, pivot), "mx_powmod_launch_form")
        return waves.value, pivot.value

    def set_segments(self, segments: int) -> None:
        """Launches one mx_powmod_nsquare_run exponentiation is cut into (0 = automatic, 1..64)."""
        if not 0 <= segments <= 64:
            raise ValueError("segments must be 0..64")
        self._segments = int(segments)

    def set_fixed_window(self, enable: bool) -> None:
        """Partial decryptions (powmod_nsquare_*) with a FIXED-window tape (MX_PLAN_FIXED_WINDOW, include/mxpaillier.h): the
        number and order of the squarings and multiplications of a launch then depend on the exponent's bit length
        only — the exponent is the party's secret share folded with its Lagrange coefficient (PSK:79-85), and the default
        sliding-window tape is a function of its bits, like gmpy2's mpz_powm.  +3.6 % instructions at key_length 2048 (and twice
        the window table);
        the table row a window reads is still chosen by the secret digit.  Same results bit for bit."""
        self._fixed_window = bool(enable)

    def set_priority_aux(self, enable: bool) -> None:
        """With several launches in flight on several streams, the small kernels of a step (recombination,
        verdict, Jacobi filter, selection: 0.5-5 ms of work) queue for wavefront slots behind the other streams'
        full-machine modexp launches — 4-34 ms of waiting measured (profiles/r02_bench_*_rocprof_summary.txt).
        When enabled they run on a HIGH-PRIORITY companion of the calling stream, ordered by events on both sides
        (the caller still sees them in stream order): the dispatcher hands them the first slots that free up."""
        self._priority_aux = bool(enable)

    def _small(self, fn):
        """fn() on the current stream, or — set_priority_aux — on its high-priority companion between two waits."""
        if not self._priority_aux:
            return fn()
        torch = self.torch
        cur = torch.cuda.current_stream(self.device)
        key = int(cur.cuda_stream)
        aux = self._aux.get(key)
        if aux is None:
            with torch.cuda.device(self.device):
                aux = self._aux[key] = torch.cuda.Stream(device=self.device, priority=-1)
        aux.wait_stream(cur)
        with torch.cuda.stream(aux):
            out = fn()
        cur.wait_stream(aux)
        for t in (out if isinstance(out, tuple) else (out,)):
            if hasattr(t, "record_stream"):
                t.record_stream(cur)         # allocated on the companion, used by the caller's stream from here on
        return out

    def selftest_lanes(self) -> int:
        with self.torch.cuda.device(self.device):
            return _lib.check(self.lib.mx_selftest_lanes(self._stream_ptr()), "mx_selftest_lanes")

    def geometry(self, mod_bits: int, batch: int = 1, groups: int = 1) -> Tuple[int, int, int, int]:
        """(lanes per element, limbs per lane, limb bits, blocks) of a generic-modulus modexp launch."""
        import ctypes

        k, l, w, b = (ctypes.c_int() for _ in range(4))
        _lib.check(self.lib.mx_powmod_geometry_for(mod_bits, batch, groups, self._lpl_generic(mod_bits), k, l, w, b), "mx_powmod_geometry_for")
        return k.value, l.value, w.value, b.value

    def debug_knob(self, knob: str, value: int) -> None:
        """Developer overrides of the library (include/mxpaillier.h: mx_debug_knob; process-wide, 0 restores
        the default): "n2_segments", "jacobi_max_batches", "n2_timeslice" (1 never, 2 always), "n2_friendly_1w" (1 never)."""
        ids = {"n2_segments": 1, "jacobi_max_batches": 2, "n2_timeslice": 3, "n2_friendly_1w": 4, "generic_latency": 5, "n2_split": 6, "bi_pivot": 7, "lat_lanes": 8, "n2_bipair": 9}
        _lib.check(self.lib.mx_debug_knob(ids[knob], int(value)), "mx_debug_knob")

    def cu_slice_streams(self, n: int) -> List[Any]:
        """`n` streams (2..8) whose kernels are confined to disjoint slices of the compute units — the same CU range
        in every XCD (mx_stream_create_cu_slice) — as torch streams.  For callers that keep several SMALL launches in
        flight (each fitting its slice at about one wavefront per SIMD): on ordinary streams the dispatcher stacks
        them on the same CUs.  Created once per engine and n; they live as long as the engine."""
        import ctypes

        cache = self.__dict__.setdefault("_cu_streams", {})
        if n not in cache:
            out = []
            with self.torch.cuda.device(self.device):
                for k in range(n):
                    ptr = ctypes.c_void_p()
                    _lib.check(self.lib.mx_stream_create_cu_slice(k, n, 0, ctypes.byref(ptr)), "mx_stream_create_cu_slice")
                    out.append(self.torch.cuda.ExternalStream(ptr.value, device=self.device))
            cache[n] = out
        return cache[n]

    def clock_probe_start(self, microseconds: int = 300):
        """Enqueue a shader-clock probe (mx_clock_probe) on a high-priority stream of its own, so that it runs
        beside whatever the other streams have in flight; returns a handle for clock_probe_mhz."""
        torch = self.torch
        if getattr(self, "_probe_stream", None) is None:
            with torch.cuda.device(self.device):
                self._probe_stream = torch.cuda.Stream(device=self.device, priority=-1)
        with torch.cuda.device(self.device), torch.cuda.stream(self._probe_stream):
            # allocated and cleared ON the probe stream: a fill enqueued on the caller's (busy) stream would run
            # after the probe and wipe its result
            out = torch.zeros(2, dtype=torch.int64, device=self.device)
            _lib.check(self.lib.mx_clock_probe(int(microseconds), out.data_ptr(), int(self._probe_stream.cuda_stream)), "mx_clock_probe")
        return out

    def clock_probe_mhz(self, handle) -> float:
        """Shader clock in MHz measured by a finished probe (waits for it)."""
        self._probe_stream.synchronize()
        ticks = handle.cpu().tolist()
        return ticks[0] / ticks[1] * 100.0 if ticks[1] else 0.0

    def profile(self, enable: bool) -> None:
        """Start/stop recording events around every modexp kernel launch (process-wide)."""
        _lib.check(self.lib.mx_profile(1 if enable else 0), "mx_profile")

    def profile_collect(self) -> Tuple[float, int]:
        """(sum of kernel durations in ms, launches) since the last collect; waits for the launches."""
        import ctypes

        total, n = ctypes.c_double(), ctypes.c_int()
        _lib.check(self.lib.mx_profile_collect(total, n), "mx_profile_collect")
        return total.value, n.value

    def nsquare_geometry(self, n_bits: int, batch: int) -> Tuple[int, int, int, int]:
        """(lanes per element, limbs per lane, limb bits, blocks) of a powmod_nsquare launch."""
        import ctypes

        return self.nsquare_launch_shape(n_bits, batch)[:4]

    def nsquare_launch_shape(self, n_bits: int, batch: int) -> Tuple[int, int, int, int, int]:
        """(lanes per element, limbs per lane, limb bits, blocks, wavefronts per group) of a
        powmod_nsquare launch of `batch` elements with this engine's settings."""
        import ctypes

        k, l, w, b, wv = (ctypes.c_int() for _ in range(5))
        _lib.check(self.lib.mx_nsquare_launch_shape(n_bits, batch, self._lpl_n2(), self._wpg, k, l, w, b, wv), "mx_nsquare_launch_shape")
        return k.value, l.value, w.value, b.value, wv.value

    def nsquare_latency_form(self, n_bits: int) -> Optional[Tuple[int, int, int, int]]:
        """(lanes per element, data positions, pivot, largest batch the library takes the form for by itself) of the
        five-wavefront latency form of powmod_nsquare for moduli of n_bits bits, or None where it has no instance."""
        import ctypes

        k, pd, pivot, most = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        rc = self.lib.mx_nsquare_latency_form(n_bits, k, pd, pivot, most)
        if rc == -2:                                  # MX_ERR_SIZE
            return None
        _lib.check(rc, "mx_nsquare_latency_form")
        return k.value, pd.value, pivot.value, most.value

    def saturating_shape(self, n_bits: int, total: int) -> Tuple[int, int]:
        """(limbs per lane, wavefronts per group) for powmod_nsquare launches that run SIDE BY SIDE on several
        streams, `total` elements between them.  When they bring about two wavefronts per SIMD or more at one
        wavefront per group of elements and 18 limbs per lane, that shape: fewest instructions per ciphertext
        (include/mxpaillier.h: "callers that keep several launches in flight ... should pass 18 / 1").  Otherwise
        the library's choice for ONE launch of the total, which is what the launches then resemble.  The engine's
        explicit settings win."""
        import ctypes

        k, l, w, b, wv = (ctypes.c_int() for _ in range(5))
        if not (self._lpl and self._wpg) and self.lib.mx_nsquare_launch_shape(n_bits, total, self._lpl_n2() or 18, self._wpg or 1, k, l, w, b, wv) == 0:
            simds = 4 * self.torch.cuda.get_device_properties(self.device).multi_processor_count
            if wv.value == 1 and l.value == 18 and total * k.value // 64 >= 15 * simds // 8:
                return (18, 1)
        # otherwise what one PLAIN launch of the total would run (a time-sliced form is a lone launch's: 4 x 2500 ciphertexts
        # at key_length 2048 reach 214 k/s at 9 limbs per lane, 179 k/s in the shape of the time-sliced choice for 10 000)
        lo, wo = ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.mx_nsquare_pieces_shape(n_bits, total, self._lpl_n2(), self._wpg, lo, wo), "mx_nsquare_pieces_shape")
        l_, w_ = lo.value, wo.value
        if w_ == 1 and not self._wpg:
            # One launch of the total would run one wavefront per group (a few per cent ahead of two once it has a
            # wavefront for every SIMD), but the launches are `total` in PIECES: a piece with fewer wavefronts than SIMDs
            # is stacked on the CUs of its neighbours, and the two-wavefront form has twice the wavefronts to spread
            # (key_length 4096, 8 x 1024 in flight: 28.7 ms per step on two wavefronts per group, 42.2 on one)
            if self.lib.mx_nsquare_pieces_shape(n_bits, total, self._lpl_n2(), 2, lo, wo) == 0:
                return (self._lpl_n2() or lo.value, 2)
        return (self._lpl_n2() or l_, self._wpg or w_)

    def nsquare_launch_split(self, n_bits: int, batch: int) -> Optional[Tuple[int, Tuple[int, int], Tuple[int, int]]]:
        """(rows of the first launch, its shape, the shape of the rest) when ONE powmod_nsquare batch of this size is
        better run as two launches side by side (mx_nsquare_launch_split) and this engine's settings leave the choice to
        the library; None otherwise.  powmod_nsquare_t follows the hint unless the caller fixed the shape or asked for
        more than one segment (both parts run as single launches); with profile() on, the two launches count as two."""
        import ctypes

        if self._lpl or self._wpg:
            return None
        first = ctypes.c_int64()
        a, b, c, d = (ctypes.c_int() for _ in range(4))
        _lib.check(self.lib.mx_nsquare_launch_split(n_bits, batch, first, a, b, c, d), "mx_nsquare_launch_split")
        return (int(first.value), (a.value, b.value), (c.value, d.value)) if first.value else None

    def nsquare_launch_timesliced(self, n_bits: int, batch: int) -> Tuple[int, int]:
        """(resident workgroups per CU, units per group) when a powmod_nsquare launch of `batch` elements with this
        engine's settings runs in the time-sliced form (mx_nsquare_launch_timesliced), (0, 0) for a plain launch."""
        import ctypes

        r, u = ctypes.c_int(), ctypes.c_int()
        _lib.check(self.lib.mx_nsquare_launch_timesliced(n_bits, batch, self._lpl_n2(), self._wpg, r, u), "mx_nsquare_launch_timesliced")
        return r.value, u.value

    def _mods_operand(self, mods, limbs: int, odd_only: bool = True):
        """Moduli of a per-group launch as (device rows [groups, limbs], max bits).  `mods` is a sequence
        of Python ints (validated and uploaded here) or an already device-resident pair (rows, bits)."""
        if isinstance(mods, tuple) and len(mods) == 2 and hasattr(mods[0], "data_ptr"):
            rows_t, bits = mods
            if rows_t.shape[1] != limbs:
                raise ValueError("device moduli must have the row width of the operands")
            return rows_t, int(bits)
        for m in mods:
            if odd_only:
                _check_modulus(m)
        bits = _limbs.max_bits(mods)
        if bits > 32 * limbs:
            raise ValueError("modulus wider than the limb rows")
        return self.to_device(_limbs.pack(mods, limbs)), bits

    # ------------------------------------------------------------------ modexp, tensor level
    @_int_args
    def powmod_shared_t(self, bases_t, mod: int, exp: int, out_t=None):
        """out[e] = bases[e]^exp mod `mod`; bases_t: int32 [batch, limbs] on this device."""
        if exp < 0:
            raise ValueError("negative exponent: invert the base first (paillier_shared_key.py:89-91)")
        batch, limbs = bases_t.shape
        if _limbs.limbs_for(mod) > limbs:
            raise ValueError("modulus wider than the limb rows")
        elimbs = _limbs.limbs_for(exp)
        h_mod = _limbs.pack_one(mod, limbs)
        h_exp = _limbs.pack_one(exp, elimbs)
        if out_t is None:
            out_t = self.torch.empty_like(bases_t)
        with self.torch.cuda.device(self.device):
            ws = self._workspace(self.lib.mx_powmod_workspace_bytes(limbs, elimbs, batch, 1))
            rc = self.lib.mx_powmod_shared_lpl(
                bases_t.data_ptr(), out_t.data_ptr(), h_mod.ctypes.data, h_exp.ctypes.data,
                limbs, elimbs, batch, self._lpl_generic(mod.bit_length()), ws.data_ptr(), ws.numel(), self._stream_ptr(),
            )
        _lib.check(rc, "mx_powmod_shared_lpl")
        return out_t

    def powmod_multi_t(self, bases_t, mods, exps, group_size: int, out_t=None):
        """out[g*group_size+k] = bases[g*group_size+k]^exps[g] mod mods[g].  `mods` / `exps`: sequences of
        ints, or device-resident (rows, max bits) pairs — the operands reach the kernel as device rows
        either way (mx_powmod_multi_dev), so thousands of candidates cost no host-side staging."""
        batch, limbs = bases_t.shape
        mods_t, mod_bits = self._mods_operand(mods, limbs)
        groups = mods_t.shape[0]
        if isinstance(exps, tuple) and len(exps) == 2 and hasattr(exps[0], "data_ptr"):
            exps_t, exp_bits = exps[0], int(exps[1])
        else:
            if len(exps) != groups:
                raise ValueError("one exponent per modulus expected")
            if any(e < 0 for e in exps):
                raise ValueError("negative exponent")
            exp_bits = _limbs.max_bits(exps)
            exps_t = self.to_device(_limbs.pack(exps, _limbs.limbs_for_bits(exp_bits)))
        if exps_t.shape[0] != groups:
            raise ValueError("one exponent per modulus expected")
        elimbs = exps_t.shape[1]
        if batch != groups * group_size:
            raise ValueError("bases must hold groups*group_size rows")
        if out_t is None:
            out_t = self.torch.empty_like(bases_t)
        with self.torch.cuda.device(self.device):
            ws = self._workspace(self.lib.mx_powmod_workspace_bytes(limbs, elimbs, batch, groups))
            rc = self.lib.mx_powmod_multi_dev(
                bases_t.data_ptr(), out_t.data_ptr(), mods_t.data_ptr(), exps_t.data_ptr(),
                limbs, elimbs, mod_bits, exp_bits, groups, group_size, self._lpl_generic(mod_bits), ws.data_ptr(), ws.numel(),
                self._stream_ptr(),
            )
        _lib.check(rc, "mx_powmod_multi_dev")
        return out_t

    # ------------------------------------------------------------------ per-key plans
    @_int_args
    def nsquare_plan(self, n: int, exp: int) -> _Plan:
        """The plan of `x -> x^exp mod n^2` (constants and tape of mx_powmod_nsquare_prepare), cached:
        (n, exp) is a key's public modulus and the party's Lagrange-folded share (PSK:46, PSK:79-85)."""
        key = (n, exp, self._fixed_window)
        plan = self._n2_plans.get(key)
        if plan is not None:
            self._n2_plans.move_to_end(key)
            return plan
        if exp < 0:
            raise ValueError("negative exponent: invert the base first (paillier_shared_key.py:89-91)")
        _check_modulus(n)
        limbs_n = _limbs.limbs_for(n)
        elimbs = _limbs.limbs_for(exp)
        h_n = _limbs.pack_one(n, limbs_n)
        h_exp = _limbs.pack_one(exp, elimbs)
        desc = _lib.NsquarePlan()
        with self.torch.cuda.device(self.device):
            nbytes = _lib.check(self.lib.mx_nsquare_plan_bytes(limbs_n, elimbs), "mx_nsquare_plan_bytes")
            block = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
            rc = self.lib.mx_powmod_nsquare_prepare_ex(
                desc, h_n.ctypes.data, h_exp.ctypes.data, limbs_n, elimbs, _lib.MX_PLAN_FIXED_WINDOW if self._fixed_window else 0,
                block.data_ptr(), block.numel(), self._stream_ptr(),
            )
            _lib.check(rc, "mx_powmod_nsquare_prepare_ex")
            ready = self.torch.cuda.Event()
            ready.record(self.torch.cuda.current_stream(self.device))
        plan = _Plan(desc, block, self._stream_ptr(), ready)
        return self._cache_plan(self._n2_plans, key, plan)

    @_int_args
    def combine_plan(self, n: int, theta_inv: int, limbs2: int) -> _Plan:
        """The plan of the share recombination for a key (mx_combine_prepare), cached."""
        key = (n, theta_inv, limbs2)
        plan = self._combine_plans.get(key)
        if plan is not None:
            self._combine_plans.move_to_end(key)
            return plan
        _check_modulus(n)
        limbs = _limbs.limbs_for(n)
        if _limbs.limbs_for(n * n) > limbs2:
            raise ValueError("partial rows narrower than N^2")
        if not 0 <= theta_inv < n:
            raise ValueError("theta_inv must be a residue modulo N")
        h_n = _limbs.pack_one(n, limbs)
        h_t = _limbs.pack_one(theta_inv, limbs)
        desc = _lib.CombinePlan()
        with self.torch.cuda.device(self.device):
            nbytes = _lib.check(self.lib.mx_combine_plan_bytes(limbs, limbs2), "mx_combine_plan_bytes")
            block = self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)
            rc = self.lib.mx_combine_prepare(
                desc, h_n.ctypes.data, h_t.ctypes.data, limbs, limbs2, block.data_ptr(), block.numel(), self._stream_ptr()
            )
            _lib.check(rc, "mx_combine_prepare")
            ready = self.torch.cuda.Event()
            ready.record(self.torch.cuda.current_stream(self.device))
        plan = _Plan(desc, block, self._stream_ptr(), ready)
        return self._cache_plan(self._combine_plans, key, plan)

    @_int_args
    def powmod_nsquare_t(self, bases_t, n: int, exp: int, out_t=None, segments: Optional[int] = None,
                         shape: Optional[Tuple[int, int]] = None):
        """out[e] = bases[e]^exp mod n^2 (rows of the width of n^2), computed through pairs modulo n
        (include/mxpaillier.h: mx_powmod_nsquare_prepare / _run) — the fast path of the partial
        decryption PSK:92.  The per-key plan is prepared on first use; afterwards a call is launches only
        (`segments` of them, default: the engine's setting, 0 = the library's choice).  `shape` =
        (limbs per lane, wavefronts per group) overrides the engine's settings for this call."""
        if exp < 0:
            raise ValueError("negative exponent: invert the base first (paillier_shared_key.py:89-91)")
        batch, limbs2 = bases_t.shape
        _check_modulus(n)
        if _limbs.limbs_for(n * n) > limbs2:
            raise ValueError("rows narrower than N^2")
        plan = self.nsquare_plan(n, exp)
        if out_t is None:
            out_t = self.torch.empty_like(bases_t)
        # (the split form runs both parts as single launches: an explicit number of segments — the `segments` argument
        # or Engine.set_segments() above 1 — only has a meaning in the one-launch form and disables the split)
        split = None
        if shape is None and segments in (None, 0, 1) and self._segments in (0, 1) and bases_t.is_contiguous() and out_t.is_contiguous():
            split = self.nsquare_launch_split(n.bit_length(), batch)
        if split is not None:
            # one batch just above a capacity step of the wide two-wavefront shape: the part that fills the CUs once in
            # that shape on this stream, the rest at 9 limbs per lane on a companion stream at the same time
            first, shape_a, shape_b = split
            torch = self.torch
            cur = torch.cuda.current_stream(self.device)
            side = self._split_streams.get(int(cur.cuda_stream))
            if side is None:
                with torch.cuda.device(self.device):
                    side = self._split_streams[int(cur.cuda_stream)] = torch.cuda.Stream(device=self.device)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                self.powmod_nsquare_t(bases_t[first:], n, exp, out_t=out_t[first:], segments=1, shape=shape_b)
            self.powmod_nsquare_t(bases_t[:first], n, exp, out_t=out_t[:first], segments=1, shape=shape_a)
            cur.wait_stream(side)
            return out_t
        with self.torch.cuda.device(self.device):
            self._use_plan(plan)
            ws = self._workspace(self.lib.mx_powmod_nsquare_run_workspace_bytes(plan.desc, batch))
            rc = self.lib.mx_powmod_nsquare_run(
                plan.desc, bases_t.data_ptr(), out_t.data_ptr(), limbs2, batch,
                self._lpl_n2() if shape is None else int(shape[0]), self._wpg if shape is None else int(shape[1]),
                self._segments if segments is None else int(segments), ws.data_ptr(), ws.numel(), self._stream_ptr(),
            )
        _lib.check(rc, "mx_powmod_nsquare_run")
        return out_t

    @_int_args
    def powmod_nsquare_batch(self, bases: Sequence[int], exp: int, n: int, keep_rows: bool = False):
        """[pow_mod(b, exp, n*n) for b in bases] through the N-adic pair kernel.  Sequences of
        PIPELINE_MIN elements or more are cut into chunks that run on several streams: the chunks
        together fill the machine (one 10 000-element launch occupies 61 % of the SIMDs), and the
        packing of chunk k+1 and the PCIe copies overlap the modexps of chunk k.
        With ``keep_rows`` the result is ``(ints, rows)`` where ``rows`` is the device-resident copy of
        the results (an opaque column for ``combine_columns``: the party's own partial decryptions go
        into the recombination without being packed a second time)."""
        if len(bases) == 0:
            return ([], None) if keep_rows else []
        _check_modulus(n)
        n2 = n * n
        limbs2 = _limbs.limbs_for(n2)
        vals = bases if isinstance(bases, list) else list(bases)
        if len(vals) < self.PIPELINE_MIN:
            import time as _t

            # rows are packed straight into a page-locked buffer and come back through one (a pageable copy of the 5.7 MB
            # of 10 000 ciphertexts costs ~1 ms each way: 2 of the 46 ms of such a call)
            torch = self.torch
            count = len(vals)
            t0 = _t.perf_counter()
            in_pin, out_pin = self._pinned("in", count, limbs2), self._pinned("out", count, limbs2)
            in_np = in_pin.numpy().view(np.uint32)
            try:
                _limbs.pack_into(vals, limbs2, in_np, 0)
                _limbs.reduce_rows(in_np, n2)
            except ValueError:                        # a value that does not fit the rows, or a negative one
                in_np[:] = _limbs.pack_reduced(vals, limbs2, n2)
            t1 = _t.perf_counter()
            # a lone launch that is waited for right away: nothing else is in flight whose drain segments
            # could shorten, and the three extra segment boundaries would cost ~1 % (a time-sliced launch keeps the
            # library's number of units per group)
            lone_segments = None
            if self._segments == 0:
                lone_segments = 0 if self.nsquare_launch_timesliced(n.bit_length(), count)[0] else 1
            out_t = self.powmod_nsquare_t(in_pin.to(self.device, non_blocking=True), n, exp, segments=lone_segments)
            out_pin.copy_(out_t, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()
            t2 = _t.perf_counter()
            res = _limbs.unpack(out_pin.numpy().view(np.uint32))
            self.last_timing = {"chunks": 1, "pack_s": t1 - t0, "copies_and_gpu_s": t2 - t1, "unpack_s": _t.perf_counter() - t2}
            return (res, out_t) if keep_rows else res
        self.nsquare_plan(n, exp)          # prepared once, before the chunks fan out over streams
        kept: List[Any] = []
        # the chunks run side by side and fill the machine between them: the shape for that is not the one a lone
        # chunk would get
        chunk_shape = self.saturating_shape(n.bit_length(), len(vals))

        def launch(t):
            out_t = self.powmod_nsquare_t(t, n, exp, shape=chunk_shape)
            if keep_rows:
                kept.append(out_t)
            return out_t

        res = self._pipelined(vals, limbs2, limbs2, launch, modulus=n2)
        if not keep_rows:
            return res
        # the chunk results were allocated on the side streams; the concatenation reads them on the
        # current stream (ordered after the side streams by _pipelined): tell the allocator
        cur = self.torch.cuda.current_stream(self.device)
        for t in kept:
            t.record_stream(cur)
        return res, self.torch.cat(kept, dim=0)

    def powmod_nsquare_groups(self, jobs: Sequence[Tuple[Sequence[int], int, int]]) -> List[List[int]]:
        """[[pow_mod(b, exp, n*n) for b in bases] for (bases, exp, n) in jobs] with the jobs' launches SIDE BY SIDE on
        separate streams: the partial decryptions of several keys — or of the parties of one key that share this
        process and GPU (``distributed=False``: same N, every party its own exponent, hence its own tape) — that are
        pending at the same time (coalesce.Coalescer).  A launch of up to ~1000 ciphertexts lasts as long as one
        wavefront's dependent chain whatever its size, so k small jobs one after the other cost k chains and side by
        side one.  Every job is launched in the shape that suits the SUM in flight (``saturating_shape``)."""
        jobs = [(b if isinstance(b, list) else list(b), int(e), int(n)) for b, e, n in jobs]
        live = [k for k, (b, _, _) in enumerate(jobs) if len(b)]
        total = sum(len(jobs[k][0]) for k in live)
        if len(live) <= 1 or total >= self.PIPELINE_MIN:
            return [self.powmod_nsquare_batch(b, e, n) if len(b) else [] for b, e, n in jobs]
        torch = self.torch
        streams = self._chunk_streams(min(len(live), self.PIPELINE_STREAMS))
        cur = torch.cuda.current_stream(self.device)
        n_bits = max(jobs[k][2].bit_length() for k in live)
        shape = self.saturating_shape(n_bits, total)
        outs: Dict[int, Any] = {}
        for j, k in enumerate(live):
            bases, exp, n = jobs[k]
            _check_modulus(n)
            n2 = n * n
            limbs2 = _limbs.limbs_for(n2)
            rows = _limbs.pack_reduced(bases, limbs2, n2)
            self.nsquare_plan(n, exp)                     # prepared on the caller's stream, before the fan-out
            side = streams[j % len(streams)]
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                outs[k] = self.powmod_nsquare_t(self.to_device(rows), n, exp, segments=1, shape=shape)
        res: List[List[int]] = [[] for _ in jobs]
        for j, k in enumerate(live):
            with torch.cuda.stream(streams[j % len(streams)]):
                res[k] = _limbs.unpack(self.to_host(outs[k]))     # copied on, and waited for through, the producing stream
        for side in streams[: len(live)]:
            cur.wait_stream(side)
        return res

    # ------------------------------------------------------------------ chunked execution on several streams
    PIPELINE_MIN = 20000       # elements from which an int-level batch is cut into chunks
    PIPELINE_STREAMS = 8

    def stream_concurrency(self, streams: Sequence[Any], spin_us: int = 400) -> List[Any]:
        """The largest prefix-greedy subset of `streams` whose kernels demonstrably run side by side.
        The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the variable was
        set before the runtime initialised — it cannot be queried or changed afterwards) and two streams
        on one queue serialise.  Measured, not assumed: every candidate stream gets one idle one-wavefront
        kernel (mx_spin) together with the streams accepted so far; if the set takes about one spin it is
        concurrent, if it takes two the newcomer shares a queue with an accepted stream and is dropped.
        Waits for the device first (one-time cost when the chunk streams are created)."""
        import time as _t

        torch = self.torch
        accepted: List[Any] = []

        def spin_all(trial) -> float:
            best = None
            for _ in range(4):                           # the first launch on a fresh stream binds its queue; best of the rest
                t0 = _t.perf_counter()
                for st in trial:
                    _lib.check(self.lib.mx_spin(spin_us, int(st.cuda_stream)), "mx_spin")
                for st in trial:
                    st.synchronize()
                dt = _t.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            return best

        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            one = None                                   # what ONE spin costs here and now (launch + clock ramp included)
            for cand in streams:
                if one is None:
                    one = max(spin_all([cand]), 1e-6 * spin_us)
                    accepted.append(cand)
                    continue
                # side by side: about one spin; sharing a queue with an accepted stream: two
                if spin_all(accepted + [cand]) < 1.6 * one:
                    accepted.append(cand)
        return accepted

    RECHECK_CAPPED_EVERY = 16       # a capped engine probes again every so many long batches (the cap may have been a hiccup)

    def _chunk_streams(self, wanted: int) -> List[Any]:
        """`wanted` streams for the chunks of a long int-level batch — or fewer, if the process does not have
        that many concurrently running queues (one warning, then larger chunks on the streams that do run
        side by side).  The verdict is a wall-clock measurement on a GPU that others may be using, so a capped
        engine does not keep it forever: it measures again every RECHECK_CAPPED_EVERY requests."""
        torch = self.torch
        if self._side_streams_capped:
            self._capped_calls = getattr(self, "_capped_calls", 0) + 1
            if self._capped_calls % self.RECHECK_CAPPED_EVERY == 0:
                self._side_streams_capped = False
        if len(self._side_streams) < wanted and not self._side_streams_capped:
            with torch.cuda.device(self.device):
                # high-priority streams are served by their own set of hardware queues: the chunks do not
                # collide with (and serialise behind) the caller's other streams even when the process runs
                # with few hardware queues (profiles/r02_hw_queue_collisions.txt)
                pool = self.__dict__.setdefault("_side_stream_pool", list(self._side_streams))
                while len(pool) < wanted:
                    # (the runtime serves at most four high-priority streams side by side whatever GPU_MAX_HW_QUEUES says —
                    # measured: 4 of 7 with 16 queues —, so the streams beyond four are ordinary ones, which do get queues
                    # of their own when the process was started with enough of them)
                    pool.append(torch.cuda.Stream(device=self.device, priority=-1 if len(pool) < 4 else 0))
                cands = self._side_streams + [st for st in pool if st not in self._side_streams][: wanted - len(self._side_streams)]
            ok = self.stream_concurrency(cands)
            if len(ok) < len(cands):
                import warnings

                if not getattr(self, "_capped_warned", False):
                    self._capped_warned = True
                    # what the PROBE measured, and only then a guess at why: with enough hardware queues configured the
                    # shortfall is the runtime's own stream-to-queue mapping (streams of the process created earlier hold
                    # queues too), not something the caller did
                    import os

                    queues = os.environ.get("GPU_MAX_HW_QUEUES")
                    if queues is None or not queues.isdigit() or int(queues) < 8:
                        why = (f"GPU_MAX_HW_QUEUES is {queues or 'unset (runtime default: 4)'}; call protocols.distributed_keygen_amd."
                               "configure_hw_queues() before the first GPU call, or set GPU_MAX_HW_QUEUES=16")
                    else:
                        why = (f"GPU_MAX_HW_QUEUES={queues} was in effect, so this is the runtime's mapping of this process's streams "
                               "onto its hardware queues, not a missing setting; the engine probes again later")
                    warnings.warn(
                        f"protocols.distributed_keygen_amd: the concurrency probe measured {len(ok)} of {len(cands)} side streams "
                        f"running side by side ({why}): long batches are cut into {max(1, len(ok))} chunks instead of {wanted}",
                        RuntimeWarning, stacklevel=3)
                self._side_streams_capped = True
            self._side_streams = ok
        return self._side_streams[: max(1, min(wanted, len(self._side_streams)))]

    def _pinned(self, which: str, rows: int, limbs: int):
        buf = self._pin.get(which)
        if buf is None or buf.numel() < rows * limbs:
            buf = self.torch.empty(rows * limbs, dtype=self.torch.int32, pin_memory=True)
            self._pin[which] = buf
        return buf[: rows * limbs].view(rows, limbs)

    def _staged_rows(self, which: str, values, limbs: int, moduli, nested: int = 0):
        """ints -> device rows [len(values), limbs] of their residues (limbs.pack_reduced) through the page-locked buffer
        `which`: packed in place, one asynchronous copy.  The caller synchronises with the stream before it returns (every
        int-level entry point fetches a result), so the buffer is free again by the time anybody packs into it.
        `nested` = n: `values` is one list per group, every group `n` rows (its first n values, zero padded) —
        limbs.pack_nested_into, no flattened copy."""
        if nested:
            lists = values if isinstance(values, (list, tuple)) else list(values)
            count = len(lists) * nested
        else:
            vals = values if isinstance(values, (list, tuple)) else list(values)
            count = len(vals)
        pin = self._pinned(which, count, limbs)
        rows = pin.numpy().view(np.uint32)
        try:
            if nested:
                _limbs.pack_nested_into(lists, nested, limbs, rows, 0)
            else:
                _limbs.pack_into(vals, limbs, rows, 0)
            _limbs.reduce_rows(rows, moduli)
        except ValueError:                              # a value that does not fit the rows, or a negative one
            if nested:
                vals = [v for g in lists for v in (list(g)[:nested] + [0] * (nested - min(nested, len(g))))]
            rows[:] = _limbs.pack_reduced(vals, limbs, moduli)
        return pin.to(self.device, non_blocking=True)

    def _fetched_ints(self, which: str, rows_t, groups=None):
        """Device rows -> ints through the page-locked buffer `which` (waits for the current stream).  `groups` =
        (counts, stride): one list per group instead (limbs.unpack_groups)."""
        pin = self._pinned(which, rows_t.shape[0], rows_t.shape[1])
        pin.copy_(rows_t, non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()
        if groups is not None:
            return _limbs.unpack_groups(pin.numpy().view(np.uint32), groups[0], groups[1])
        return _limbs.unpack(pin.numpy().view(np.uint32))

    def _pipelined(self, vals: List[int], limbs_in: int, limbs_out: int, launch, modulus: int = 0) -> List[int]:
        """ints -> ints through `launch(device rows) -> device rows`, chunked over side streams with
        pinned staging buffers: pack chunk k+1 on the host while chunk k is copied and computed.  With
        `modulus` the packed rows are reduced modulo it (bulk: limbs.reduce_rows)."""
        import time as _t

        torch = self.torch
        total = len(vals)
        streams = self._chunk_streams(max(4, min(self.PIPELINE_STREAMS, total // 10000)))
        nchunks = len(streams)
        per = -(-total // nchunks)
        in_pin = self._pinned("in", total, limbs_in)
        out_pin = self._pinned("out", total, limbs_out)
        in_np = in_pin.numpy().view(np.uint32)
        out_np = out_pin.numpy().view(np.uint32)
        cur = torch.cuda.current_stream(self.device)
        events, bounds, keep = [], [], []
        pack_s = 0.0
        t_start = _t.perf_counter()
        for k in range(nchunks):
            lo, hi = k * per, min(total, (k + 1) * per)
            if lo >= hi:
                break
            t0 = _t.perf_counter()
            try:
                _limbs.pack_into(vals[lo:hi], limbs_in, in_np, lo)
            except ValueError:
                if not modulus:
                    raise
                _limbs.pack_into([int(v) % modulus for v in vals[lo:hi]], limbs_in, in_np, lo)
            if modulus:
                _limbs.reduce_rows(in_np[lo:hi], modulus)
            pack_s += _t.perf_counter() - t0
            side = streams[k]
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                d_in = in_pin[lo:hi].to(self.device, non_blocking=True)
                d_out = launch(d_in)
                out_pin[lo:hi].copy_(d_out, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            keep.append((d_in, d_out))
            events.append(ev)
            bounds.append((lo, hi))
        out: List[int] = []
        wait_s = unpack_s = 0.0
        for ev, (lo, hi) in zip(events, bounds):
            t0 = _t.perf_counter()
            ev.synchronize()
            t1 = _t.perf_counter()
            out.extend(_limbs.unpack(out_np[lo:hi]))
            wait_s += t1 - t0
            unpack_s += _t.perf_counter() - t1
        for side in streams[: len(events)]:
            cur.wait_stream(side)
        self.last_timing = {"chunks": len(events), "pack_s": pack_s, "wait_for_gpu_s": wait_s, "unpack_s": unpack_s,
                            "total_s": _t.perf_counter() - t_start}
        return out

    # ------------------------------------------------------------------ modexp, int level
    @_int_args
    def powmod_batch(self, bases: Sequence[int], exp: int, mod: int) -> List[int]:
        """[pow_mod(b, exp, mod) for b in bases] on the GPU (exp >= 0)."""
        if len(bases) == 0:
            return []
        _check_modulus(mod)
        limbs = _limbs.limbs_for(mod)
        rows = _limbs.pack_reduced(bases, limbs, mod)
        out = self.powmod_shared_t(self.to_device(rows), mod, exp)
        return _limbs.unpack(self.to_host(out))

    @_int_args
    def powmod_batch_multi(
        self, bases: Sequence[Sequence[int]], exps: Sequence[int], mods: Sequence[int]
    ) -> List[List[int]]:
        """[[pow_mod(b, exps[g], mods[g]) for b in bases[g]] for g]; ragged groups are padded."""
        groups = len(mods)
        if groups == 0:
            return []
        if len(bases) != groups or len(exps) != groups:
            raise ValueError("bases, exps and mods must have one entry per group")
        for m in mods:
            _check_modulus(m)
        gsize = max(len(b) for b in bases)
        if gsize == 0:
            return [[] for _ in bases]
        limbs = _limbs.limbs_for_bits(_limbs.max_bits(mods))
        flat = []
        for b in bases:
            flat.extend(b)
            flat.extend([0] * (gsize - len(b)))
        rows = _limbs.pack_reduced(flat, limbs, list(mods))
        out = self.powmod_multi_t(self.to_device(rows), list(mods), list(exps), gsize)
        vals = _limbs.unpack(self.to_host(out))
        return [vals[g * gsize : g * gsize + len(bases[g])] for g in range(groups)]


    # ------------------------------------------------------------------ modular multiplication / inversion / encryption
    @_int_args
    def mulmod_t(self, a_t, b_t, mod: int, out_t=None):
        """out[e] = a[e]*b[e] mod `mod`; int32 rows [batch, limbs]; out_t may alias an input."""
        batch, limbs = a_t.shape
        if tuple(b_t.shape) != (batch, limbs):
            raise ValueError("operands must have the same shape")
        h_mod = _limbs.pack_one(mod, limbs)
        if out_t is None:
            out_t = self.torch.empty_like(a_t)
        with self.torch.cuda.device(self.device):
            ws = self._workspace(self.lib.mx_mulmod_workspace_bytes(limbs))
            rc = self.lib.mx_mulmod_shared(
                a_t.data_ptr(), b_t.data_ptr(), out_t.data_ptr(), h_mod.ctypes.data, limbs, batch,
                ws.data_ptr(), ws.numel(), self._stream_ptr(),
            )
        _lib.check(rc, "mx_mulmod_shared")
        return out_t

    @_int_args
    def mulmod_batch(self, a: Sequence[int], b: Sequence[int], mod: int) -> List[int]:
        if len(a) != len(b):
            raise ValueError("operands must have the same length")
        if len(a) == 0:
            return []
        _check_modulus(mod)
        limbs = _limbs.limbs_for(mod)
        at = self.to_device(_limbs.pack_reduced(a, limbs, mod))
        bt = self.to_device(_limbs.pack_reduced(b, limbs, mod))
        return _limbs.unpack(self.to_host(self.mulmod_t(at, bt, mod)))

    DIRECT_MODINV_MAX = 4      # elements inverted directly (one wavefront each); longer batches use the product tree

    @_int_args
    def modinv_direct_t(self, x_t, mod: int):
        """Row-wise modular inverse on the device, one wavefront per row (mx_modinv): for the few
        values that need it (the root of the product tree, theta of a key).  Raises ValueError (like
        ``pow(v, -1, m)``) if some row is not invertible."""
        batch, limbs = x_t.shape
        _check_modulus(mod)
        h_mod = _limbs.pack_one(mod, limbs)
        out_t = self.torch.empty_like(x_t)
        status_t = self.torch.empty(batch, dtype=self.torch.uint8, device=self.device)
        with self.torch.cuda.device(self.device):
            ws = self._workspace(self.lib.mx_modinv_workspace_bytes(limbs))
            rc = self.lib.mx_modinv(
                x_t.data_ptr(), out_t.data_ptr(), status_t.data_ptr(), h_mod.ctypes.data, limbs, batch,
                ws.data_ptr(), ws.numel(), self._stream_ptr(),
            )
        _lib.check(rc, "mx_modinv")
        if int(status_t.sum().item()) != 0:
            raise ValueError("base is not invertible for the given modulus")
        return out_t

    @_int_args
    def modinv_t(self, x_t, mod: int):
        """Row-wise modular inverse by Montgomery's trick as a product tree on the device: 3 modular
        multiplications per element in ~2 log2(batch) launches, and the root inverted on the device too
        (mx_modinv) — no host big-integer step (PSK:89-91 over a batch).
        Raises ValueError (like ``pow(v, -1, m)``) if some row is not invertible."""
        torch = self.torch
        levels = [x_t.contiguous()]
        while levels[-1].shape[0] > self.DIRECT_MODINV_MAX:
            cur = levels[-1]
            m = cur.shape[0]
            prod = self.mulmod_t(cur[0 : m - (m & 1) : 2].contiguous(), cur[1:m:2].contiguous(), mod)
            if m & 1:
                prod = torch.cat([prod, cur[m - 1 : m]], dim=0)
            levels.append(prod)
        inv = self.modinv_direct_t(levels[-1], mod)        # ValueError if not invertible
        for cur in reversed(levels[:-1]):
            m = cur.shape[0]
            half = m // 2
            left, right = cur[0 : 2 * half : 2].contiguous(), cur[1 : 2 * half : 2].contiguous()
            nxt = torch.empty_like(cur)
            nxt[0 : 2 * half : 2] = self.mulmod_t(inv[:half].contiguous(), right, mod)
            nxt[1 : 2 * half : 2] = self.mulmod_t(inv[:half].contiguous(), left, mod)
            if m & 1:
                nxt[m - 1] = inv[half]
            inv = nxt
        return inv

    @_int_args
    def modinv_batch(self, values: Sequence[int], mod: int) -> List[int]:
        """[mod_inv(v, mod) for v in values] (PSK:90 over a batch)."""
        if len(values) == 0:
            return []
        _check_modulus(mod)
        limbs = _limbs.limbs_for(mod)
        x_t = self.to_device(_limbs.pack_reduced(values, limbs, mod))
        return _limbs.unpack(self.to_host(self.modinv_t(x_t, mod)))

    @_int_args
    def encrypt_batch(self, messages: Sequence[int], randomness: Sequence[int], n: int) -> List[int]:
        """Paillier encryption with g = n + 1:  c = (1 + m n) * r^n mod n^2 for every (m, r)."""
        if len(messages) != len(randomness):
            raise ValueError("one randomness per message expected")
        if len(messages) == 0:
            return []
        _check_modulus(n)
        n2 = n * n
        limbs = _limbs.limbs_for(n2)
        rn_t = self.powmod_nsquare_t(self.to_device(_limbs.pack_reduced(randomness, limbs, n2)), n, n)
        g_t = self.to_device(_limbs.pack([(1 + (m % n) * n) % n2 for m in messages], limbs))
        return _limbs.unpack(self.to_host(self.mulmod_t(rn_t, g_t, n2, out_t=rn_t)))

    @_int_args
    def randomize_batch(self, ciphertexts: Sequence[int], randomness: Sequence[int], n: int) -> List[int]:
        """Re-randomisation of Paillier ciphertexts, c * r^n mod n^2 for every (c, r) — what the
        un-vendored scheme's ``randomize`` does before a ciphertext is sent (README.md:165-171 of the
        reference: a ciphertext must be fresh when it leaves); r^n through the N^2 pair kernel."""
        if len(ciphertexts) != len(randomness):
            raise ValueError("one randomness per ciphertext expected")
        if len(ciphertexts) == 0:
            return []
        _check_modulus(n)
        n2 = n * n
        limbs = _limbs.limbs_for(n2)
        rn_t = self.powmod_nsquare_t(self.to_device(_limbs.pack_reduced(randomness, limbs, n2)), n, n)
        c_t = self.to_device(_limbs.pack_reduced(ciphertexts, limbs, n2))
        return _limbs.unpack(self.to_host(self.mulmod_t(rn_t, c_t, n2, out_t=rn_t)))

    # ------------------------------------------------------------------ Shamir field of the key generation
    @_int_args
    def shamir_fma_t(self, a_t, b_t, c_t, prime: int, out_t=None):
        """out[e] = (a[e]*b[e] + c[e]) mod prime — this party's share of every candidate modulus
        (`p * q` then `+= zero`, DK:1274-1277); int32 rows [batch, limbs]."""
        batch, limbs = a_t.shape
        if tuple(b_t.shape) != (batch, limbs) or tuple(c_t.shape) != (batch, limbs):
            raise ValueError("operands must have the same shape")
        _check_modulus(prime)
        h_mod = _limbs.pack_one(prime, limbs)
        if out_t is None:
            out_t = self.torch.empty_like(a_t)
        with self.torch.cuda.device(self.device):
            ws = self._workspace(self.lib.mx_field_workspace_bytes(limbs, 0))
            rc = self.lib.mx_fma_mod(
                a_t.data_ptr(), b_t.data_ptr(), c_t.data_ptr(), out_t.data_ptr(), h_mod.ctypes.data, limbs, batch,
                ws.data_ptr(), ws.numel(), self._stream_ptr(),
            )
        _lib.check(rc, "mx_fma_mod")
        return out_t

    @_int_args
    def shamir_lincomb_t(self, x_t, coeffs: Sequence[int], prime: int, out_t=None):
        """out[e] = sum_t coeffs[t] * x[t][e] mod prime; x_t int32 [terms, batch, limbs].  With the
        Lagrange coefficients at 0 this is `candidate_n.reconstruct()` (DK:1284) for a whole round; the
        result rows are the candidate moduli, ready for sieve_t / biprime_v_t on the device."""
        terms, batch, limbs = x_t.shape
        if len(coeffs) != terms:
            raise ValueError("one coefficient per term expected")
        _check_modulus(prime)
        h_mod = _limbs.pack_one(prime, limbs)
        h_cf = _limbs.pack([c % prime for c in coeffs], limbs)
        if out_t is None:
            out_t = self.torch.empty((batch, limbs), dtype=self.torch.int32, device=self.device)
        with self.torch.cuda.device(self.device):
            ws = self._workspace(self.lib.mx_field_workspace_bytes(limbs, terms))
            rc = self.lib.mx_lincomb_mod(
                x_t.data_ptr(), h_cf.ctypes.data, out_t.data_ptr(), h_mod.ctypes.data, limbs, terms, batch,
                ws.data_ptr(), ws.numel(), self._stream_ptr(),
            )
        _lib.check(rc, "mx_lincomb_mod")
        return out_t

    @_int_args
    def shamir_fma_batch(self, a: Sequence[int], b: Sequence[int], c: Sequence[int], prime: int) -> List[int]:
        if not (len(a) == len(b) == len(c)):
            raise ValueError("operands must have the same length")
        if len(a) == 0:
            return []
        limbs = _limbs.limbs_for(prime)
        ts = [self.to_device(_limbs.pack_reduced(col, limbs, prime)) for col in (a, b, c)]
        return _limbs.unpack(self.to_host(self.shamir_fma_t(ts[0], ts[1], ts[2], prime)))

    @_int_args
    def shamir_lincomb_batch(self, columns: Sequence[Sequence[int]], coeffs: Sequence[int], prime: int) -> List[int]:
        """[sum_t coeffs[t] * columns[t][e] mod prime for e]; one column per term."""
        if len(columns) == 0 or len(columns[0]) == 0:
            return []
        if any(len(c) != len(columns[0]) for c in columns):
            raise ValueError("columns must have the same length")
        limbs = _limbs.limbs_for(prime)
        x = np.stack([_limbs.pack_reduced(col, limbs, prime) for col in columns])
        return _limbs.unpack(self.to_host(self.shamir_lincomb_t(self.to_device(x), coeffs, prime)))

    @_int_args
    def shamir_reconstruct_sieve_batch(self, columns: Sequence[Sequence[int]], coeffs: Sequence[int], prime: int,
                                       primes: Sequence[int], keep_rows: bool = False):
        """The candidate moduli of a round and their small-prime verdicts in one device pass
        (DK:1284 + DK:1288-1292): reconstruction rows go straight into the sieve; only the verdict bytes
        and the moduli of the SURVIVORS (~2 % of a round) come back to the host.
        Returns (has_small_divisor per candidate, {candidate index: modulus} for the survivors); with
        ``keep_rows`` a third value: the survivors' moduli as device rows (in index order, an opaque
        handle for ``biprime_v_batch`` / ``biprime_verdict_columns``: the round's later steps take their moduli
        from it instead of packing them again), None if nothing survived."""
        if len(columns) == 0 or len(columns[0]) == 0:
            return ([], {}, None) if keep_rows else ([], {})
        limbs = _limbs.limbs_for(prime)
        # the parties' columns are packed side by side into ONE page-locked buffer (no per-column array, no np.stack of
        # 5 x 65 536 x 2100-bit shares, no pageable copy: 38 -> 28 ms of host time per 65 536-candidate round)
        ncols, count = len(columns), len(columns[0])
        if any(len(col) != count for col in columns):
            raise ValueError("columns must have the same length")
        pin = self._pinned("shares", ncols * count, limbs)
        rows = pin.numpy().view(np.uint32)
        for i, col in enumerate(columns):
            try:
                _limbs.pack_into(col if isinstance(col, (list, tuple)) else list(col), limbs, rows, i * count)
            except ValueError:                      # a share that does not fit the field's rows, or a negative one
                rows[i * count : (i + 1) * count] = _limbs.pack_reduced(col, limbs, prime)
        _limbs.reduce_rows(rows, prime)
        x_t = pin.to(self.device, non_blocking=True).view(ncols, count, limbs)
        mods_t = self.shamir_lincomb_t(x_t, coeffs, prime)
        primes = [int(q) for q in primes]
        if len(primes) == 0:
            bad = np.zeros(mods_t.shape[0], dtype=np.uint8)
        else:
            bad = self.sieve_t(mods_t, primes).cpu().numpy()
        keep = np.nonzero(bad == 0)[0]
        survivors: Dict[int, int] = {}
        rows = None
        if len(keep):
            idx = self.torch.from_numpy(keep.astype(np.int64)).to(self.device)
            rows_t = mods_t.index_select(0, idx)
            vals = _limbs.unpack(self.to_host(rows_t))
            survivors = {int(k): v for k, v in zip(keep, vals)}
            rows = _ModulusRows(rows_t, _limbs.max_bits(vals))
        flags = bad.astype(bool).tolist()
        return (flags, survivors, rows) if keep_rows else (flags, survivors)

    # ------------------------------------------------------------------ Jacobi symbol
    def jacobi_t(self, values_t, mods, group_size: int, out_t=None, first: int = 0, count: Optional[int] = None,
                 skip_counts_t=None, skip_threshold: int = 0):
        """int8 [groups*group_size]: Jacobi symbol (values[g*group_size+k] / mods[g]) (DK:1089).
        `mods`: sequence of ints or a device-resident (rows, max bits) pair.  With first / count only
        the rows [first, first+count) of every group are evaluated (the other entries of out_t are left
        as they are); groups with skip_counts_t[g] >= skip_threshold are skipped."""
        total, limbs = values_t.shape
        if not (isinstance(mods, tuple) and hasattr(mods[0], "data_ptr")):
            for m in mods:
                if m < 1 or m % 2 == 0:
                    raise ValueError("n should be an odd positive integer")
        mods_t, _ = self._mods_operand(mods, limbs, odd_only=False)
        groups = mods_t.shape[0]
        if total != groups * group_size:
            raise ValueError("values must hold groups*group_size rows")
        if count is None:
            count = group_size - first
        if out_t is None:
            full = first == 0 and count == group_size and skip_counts_t is None
            out_t = (self.torch.empty if full else self.torch.zeros)(total, dtype=self.torch.int8, device=self.device)
        with self.torch.cuda.device(self.device):
            rc = self.lib.mx_jacobi_dev_range(
                values_t.data_ptr(), out_t.data_ptr(), mods_t.data_ptr(), limbs, groups, group_size, first, count,
                skip_counts_t.data_ptr() if skip_counts_t is not None else None, skip_threshold, self._stream_ptr(),
            )
        _lib.check(rc, "mx_jacobi_dev_range")
        return out_t

    @_int_args
    def jacobi_batch(self, values: Sequence[Sequence[int]], mods: Sequence[int]) -> List[List[int]]:
        """[[jacobi_symbol(v, mods[g]) for v in values[g]] for g]; ragged groups are padded."""
        groups = len(mods)
        if groups == 0:
            return []
        for m in mods:
            if m < 1 or m % 2 == 0:
                raise ValueError("n should be an odd positive integer")
        gsize = max(len(v) for v in values)
        if gsize == 0:
            return [[] for _ in values]
        limbs = _limbs.limbs_for_bits(_limbs.max_bits(mods))
        flat = []
        for vs in values:
            flat.extend(vs)
            flat.extend([0] * (gsize - len(vs)))
        out = self.jacobi_t(self.to_device(_limbs.pack_reduced(flat, limbs, list(mods))), list(mods), gsize)
        arr = out.cpu().numpy()
        return [[int(x) for x in arr[g * gsize : g * gsize + len(values[g])]] for g in range(groups)]

    def select_first_t(self, rows_t, flags_t, group_size: int, keep: int):
        """rows_t int32 [groups*group_size, limbs], flags_t int8 [groups*group_size] -> (int32
        [groups*keep, limbs] with the first `keep` flag==1 rows of every group, int32 counts [groups])."""
        total, limbs = rows_t.shape
        groups = total // group_size
        out_t = self.torch.empty((groups * keep, limbs), dtype=self.torch.int32, device=self.device)
        cnt_t = self.torch.empty(groups, dtype=self.torch.int32, device=self.device)
        with self.torch.cuda.device(self.device):
            rc = self.lib.mx_select_first(
                rows_t.data_ptr(), flags_t.data_ptr(), out_t.data_ptr(), cnt_t.data_ptr(), limbs, groups,
                group_size, keep, self._stream_ptr(),
            )
        _lib.check(rc, "mx_select_first")
        return out_t, cnt_t

    JACOBI_ONE_LAUNCH_SYMBOLS = 16384        # 256 wavefronts: a quarter of the SIMDs

    def biprime_v_t(self, g_t, mods, exps, group_size: int, keep: int):
        """Tensor-level v-calculation of DK:1084-1099 for many candidates, nothing leaving the device:
        g_t int32 [groups*group_size, limbs] (the jointly random generators, reduced) -> (v rows int32
        [groups*keep, limbs], counts int32 [groups]): Jacobi symbols of all generators, selection of the
        first `keep` with symbol 1 (in order), v = g^exp mod N for those; rows beyond a candidate's count
        are the modexp of a zero row and are ignored by the caller."""
        limbs = g_t.shape[1]
        mods_op = self._mods_operand(mods, limbs)
        # The selection stops at `keep` generators with symbol 1 (DK:1086) and a symbol is 1 for about
        # half of them: the first 2.6 * keep generators yield `keep` ones for > 99 % of the candidates,
        # and the tail of the list is evaluated only for the candidates where they did not.
        head = min(group_size, (13 * keep + 4) // 5)
        # ... unless the launch is so small that it lasts as long as ONE thread's symbol whatever its size (a round at the
        # reference's batch sizes: 0.8 ms for 1 .. 20 candidates x 104 or x 160 symbols at key_length 2048,
        # profiles/r05_keygen_round_small.txt): a second launch would only add its latency
        if mods_op[0].shape[0] * group_size <= self.JACOBI_ONE_LAUNCH_SYMBOLS:
            head = group_size

        def filter_and_select():
            j_t = self.jacobi_t(g_t, mods_op, group_size, first=0, count=head)
            sel_t, cnt_t = self.select_first_t(g_t, j_t, group_size, keep)
            if head < group_size:
                self.jacobi_t(g_t, mods_op, group_size, out_t=j_t, first=head, count=group_size - head,
                              skip_counts_t=cnt_t, skip_threshold=keep)
                sel_t, cnt_t = self.select_first_t(g_t, j_t, group_size, keep)
            return sel_t, cnt_t

        sel_t, cnt_t = self._small(filter_and_select)
        v_t = self.powmod_multi_t(sel_t, mods_op, exps, keep)
        return v_t, cnt_t

    @_int_args
    def biprime_v_batch(
        self, g_values: Sequence[Sequence[int]], exps: Sequence[int], mods: Sequence[int], keep: int,
        mods_rows: Any = None, keep_rows: bool = False,
    ):
        """The whole v-calculation of DK:1084-1099 for many candidates on the device: Jacobi symbols of
        all generators, selection of the first `keep` with symbol 1, v = g^exp mod N for those.
        Returns per candidate the list of v values (shorter than `keep` if fewer symbols were 1).
        `mods_rows`: the handle ``shamir_reconstruct_sieve_batch(..., keep_rows=True)`` returned for exactly these
        moduli — they are then not packed and uploaded again.  With ``keep_rows`` the result is
        ``(lists, rows)`` where ``rows`` keeps this party's v values on the device for ``biprime_verdict_columns``."""
        groups = len(mods)
        if groups == 0:
            return ([], None) if keep_rows else []
        for m in mods:          # the int moduli are passed either way: an even N reconstructed from malformed shares (the
            _check_modulus(m)   # sieve's list starts at 3) must raise here, not reach the Montgomery and Jacobi kernels
        gsize = max(len(g) for g in g_values)
        if gsize == 0 or keep == 0:
            return ([[] for _ in mods], None) if keep_rows else [[] for _ in mods]
        limbs = _limbs.limbs_for_bits(_limbs.max_bits(mods))
        # one list per candidate straight into the staging buffer (short lists zero padded: symbol (0/N) = 0, never selected)
        g_t = self._staged_rows("generators", g_values, limbs, mods, nested=gsize)
        mods_op = mods_rows.operand(self, groups, limbs) if mods_rows is not None else mods
        v_t, cnt_t = self.biprime_v_t(g_t, mods_op, exps, gsize, keep)
        counts = cnt_t.cpu().numpy()
        lists = self._fetched_ints("v", v_t, groups=(counts.tolist(), keep))      # per candidate, built in one pass
        return (lists, _VRows(v_t, counts, keep)) if keep_rows else lists

    # ------------------------------------------------------------------ sieve
    @_int_args
    def sieve_t(self, cands_t, primes: Sequence[int], out_t=None):
        """uint8 [batch]: 1 iff some prime divides candidate e (distributed_keygen.py:1197-1209)."""
        batch, limbs = cands_t.shape
        h_primes = np.ascontiguousarray(np.asarray(list(primes), dtype=np.uint32))
        if out_t is None:
            out_t = self.torch.empty(batch, dtype=self.torch.uint8, device=self.device)
        with self.torch.cuda.device(self.device):
            ws = self._workspace(self.lib.mx_sieve_workspace_bytes(limbs, len(h_primes)))
            rc = self.lib.mx_sieve(
                cands_t.data_ptr(), out_t.data_ptr(), h_primes.ctypes.data, len(h_primes), limbs, batch,
                ws.data_ptr(), ws.numel(), self._stream_ptr(),
            )
        _lib.check(rc, "mx_sieve")
        return out_t

    @_int_args
    def sieve_batch(self, candidates: Sequence[int], primes: Sequence[int]) -> List[bool]:
        """[__small_prime_divisors_test(primes, n) for n in candidates]."""
        if len(candidates) == 0:
            return []
        primes = [int(p) for p in primes]
        if len(primes) == 0:
            return [False] * len(candidates)
        if any(c < 0 for c in candidates):
            raise ValueError("candidates must be non-negative")
        limbs = _limbs.limbs_for_bits(_limbs.max_bits(candidates))
        out = self.sieve_t(self.to_device(_limbs.pack(candidates, limbs)), primes)
        return [bool(x) for x in out.cpu().numpy()]

    # ------------------------------------------------------------------ share recombination
    @_int_args
    def combine_t(self, partials_t, n: int, theta_inv: int, out_t=None, status_t=None, packed: bool = False):
        """partials_t int32 [n_partials, batch, limbs2] (players 1..degree+1 in order) ->
        (plaintext rows int32 [batch, limbs(N)], status uint8 [batch], 1 = not divisible by N).
        With ``packed=True`` the result is ONE tensor [batch, limbs(N)+1] whose last word is the
        status — plaintext and status as one row (one all-gather when ciphertexts are sharded)."""
        n_partials, batch, limbs2 = partials_t.shape
        plan = self.combine_plan(n, theta_inv, limbs2)
        limbs = plan.desc.limbs
        stride = limbs + 1 if packed else limbs
        if out_t is None:
            out_t = self.torch.empty((batch, stride), dtype=self.torch.int32, device=self.device)
        if tuple(out_t.shape) != (batch, stride):
            raise ValueError("output rows of the wrong shape")
        if status_t is None and not packed:
            status_t = self.torch.empty(batch, dtype=self.torch.uint8, device=self.device)
        def run():
            with self.torch.cuda.device(self.device):
                self._use_plan(plan)
                rc = self.lib.mx_combine_run(
                    plan.desc, partials_t.data_ptr(), out_t.data_ptr(), stride,
                    status_t.data_ptr() if status_t is not None else None, n_partials, batch, self._stream_ptr(),
                )
            _lib.check(rc, "mx_combine_run")

        self._small(run)
        return out_t if packed else (out_t, status_t)

    @_int_args
    def combine_batch(
        self, partials: Sequence[Sequence[int]], n: int, theta_inv: int
    ) -> Tuple[List[int], List[bool]]:
        """partials[e] = the degree+1 partial decryptions of ciphertext e (player 1 first).
        Returns (messages, ok); ok[e] False where the reference raises ValueError (PSK:119-123)."""
        if len(partials) == 0:
            return [], []
        n_partials = len(partials[0])
        if any(len(p) != n_partials for p in partials):
            raise ValueError("every ciphertext needs the same number of partial decryptions")
        n2 = n * n
        limbs2 = _limbs.limbs_for(n2)
        rows = np.stack([_limbs.pack_reduced([p[i] for p in partials], limbs2, n2) for i in range(n_partials)])
        out_t, status_t = self.combine_t(self.to_device(rows), n, theta_inv)
        ok = [not bool(x) for x in status_t.cpu().numpy()]
        return _limbs.unpack(self.to_host(out_t)), ok

    @_int_args
    def combine_columns(self, columns: Sequence[Any], n: int, theta_inv: int) -> Tuple[List[int], List[bool]]:
        """Share recombination from one COLUMN per player (players 1..degree+1 in order): a column is the
        device rows kept by ``powmod_nsquare_batch(..., keep_rows=True)`` or a received list of partial
        decryptions (plain ints or wire-form ``{"type": "int", "data": bytes}`` entries, DK:496-505) —
        received values reach the device through codec.rows_from_wire, i.e. without a per-element
        Python conversion and without the per-ciphertext dictionaries of DK:477-505.
        Returns (messages, ok) like ``combine_batch``."""
        from . import codec

        if len(columns) == 0:
            return [], []
        _check_modulus(n)
        n2 = n * n
        limbs2 = _limbs.limbs_for(n2)
        torch = self.torch
        cols = []
        for col in columns:
            if hasattr(col, "data_ptr"):
                if col.shape[1] != limbs2:
                    raise ValueError("device column of the wrong row width")
                cols.append(col)
            else:
                cols.append(self.to_device(codec.rows_from_wire(col, limbs2, modulus=n2)))
        batch = cols[0].shape[0]
        if any(c.shape[0] != batch for c in cols):
            raise ValueError("every player's column needs one partial decryption per ciphertext")
        if batch == 0:
            return [], []
        out_t, status_t = self.combine_t(torch.stack(cols, dim=0), n, theta_inv)
        msgs = self._fetched_ints("messages", out_t)
        ok = [not bool(x) for x in status_t.cpu().numpy()]
        return msgs, ok

    # ------------------------------------------------------------------ biprimality verdict
    def biprime_verdict_t(self, v_t, mods, pass_t=None):
        """v_t int32 [n_parties, groups, n_slots, limbs] (party 1 first) -> uint8 [groups, n_slots].
        `mods`: sequence of ints or a device-resident (rows, max bits) pair."""
        n_parties, groups, n_slots, limbs = v_t.shape
        mods_t, mod_bits = self._mods_operand(mods, limbs)
        if mods_t.shape[0] != groups:
            raise ValueError("one modulus per group expected")
        if pass_t is None:
            pass_t = self.torch.empty((groups, n_slots), dtype=self.torch.uint8, device=self.device)
        def run():
            with self.torch.cuda.device(self.device):
                ws = self._workspace(self.lib.mx_verdict_workspace_bytes(limbs, n_parties, groups, n_slots))
                rc = self.lib.mx_biprime_verdict_dev(
                    v_t.data_ptr(), pass_t.data_ptr(), mods_t.data_ptr(), limbs, mod_bits, n_parties, groups, n_slots,
                    ws.data_ptr(), ws.numel(), self._stream_ptr(),
                )
            _lib.check(rc, "mx_biprime_verdict_dev")

        self._small(run)
        return pass_t

    @_int_args
    def biprime_verdict_columns(self, columns: Sequence[Any], mods: Sequence[int], n_slots: int, mods_rows: Any = None,
                                as_array: bool = False):
        """The slot tests DK:1147-1158 of many candidates from ONE COLUMN PER PARTY (party 1 first): a column is a flat
        list of groups * n_slots values (candidate-major; short candidates padded by the caller) — packed with one
        codec call — or the handle ``biprime_v_batch(..., keep_rows=True)`` returned for this party's own values,
        which are then taken from the device as they are.  `mods_rows` as in ``biprime_v_batch``.
        Returns per candidate the per-slot verdicts, like ``biprime_verdict_batch`` (``as_array``: as one numpy array)."""
        groups = len(mods)
        if groups == 0:
            return []
        if n_slots == 0:
            return [[] for _ in mods]
        for m in mods:
            _check_modulus(m)
        limbs = _limbs.limbs_for_bits(_limbs.max_bits(mods))
        torch = self.torch
        parts = []
        for col in columns:
            if isinstance(col, _VRows):
                parts.append(col.slots(self, groups, n_slots, limbs))
            elif isinstance(col, NestedColumn):
                if len(col.lists) != groups:
                    raise ValueError("a party's column needs one list per candidate")
                parts.append(self._staged_rows(f"column{len(parts)}", col.lists, limbs, mods, nested=n_slots).view(groups, n_slots, limbs))
            else:
                if len(col) != groups * n_slots:
                    raise ValueError("a party's column needs groups * n_slots values")
                parts.append(self._staged_rows(f"column{len(parts)}", col, limbs, mods).view(groups, n_slots, limbs))
        mods_op = mods_rows.operand(self, groups, limbs) if mods_rows is not None else mods
        pass_t = self.biprime_verdict_t(torch.stack(parts, dim=0), mods_op)
        arr = pass_t.cpu().numpy().astype(bool)
        return arr if as_array else arr.tolist()        # as_array: bool [groups, n_slots] for callers that reduce it with numpy

    @_int_args
    def biprime_verdict_batch(self, v: Sequence[Sequence[Sequence[int]]], mods: Sequence[int]) -> List[List[bool]]:
        """v[g][i][k]: share of party i+1 in test slot k of candidate g (all parties, equal slot counts).
        Returns per candidate the per-slot result of `v_1 == +-prod_{i>=2} v_i (mod N)` (DK:1147-1158)."""
        groups = len(mods)
        if groups == 0:
            return []
        n_parties = len(v[0])
        n_slots = len(v[0][0])
        if n_slots == 0:
            return [[] for _ in mods]
        for m in mods:
            _check_modulus(m)
        limbs = _limbs.limbs_for_bits(_limbs.max_bits(mods))
        rows = np.stack(
            [
                _limbs.pack_reduced([x for g in range(groups) for x in v[g][i][:n_slots]], limbs, list(mods))
                for i in range(n_parties)
            ]
        ).reshape(n_parties, groups, n_slots, limbs)
        pass_t = self.biprime_verdict_t(self.to_device(rows), list(mods))
        arr = pass_t.cpu().numpy().astype(bool)
        return [list(map(bool, arr[g])) for g in range(groups)]


def _check_modulus(mod: int) -> None:
    if mod < 3 or mod % 2 == 0:
        raise ValueError("modulus must be odd and >= 3 (Paillier moduli N and N^2 are)")


def _reduce(value: int, mod: int) -> int:
    value = int(value)
    return value if 0 <= value < mod else value % mod


_default_engine: Optional[Engine] = None


def default_engine() -> Engine:
    global _default_engine
    if _default_engine is None:
        _default_engine = Engine()
    return _default_engine
