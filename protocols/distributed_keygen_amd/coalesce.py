"""Awaitable micro-batcher: concurrent single-ciphertext calls of one event loop share ONE launch.

The reference's API produces lone calls from many coroutines: ``asyncio.gather(*(scheme.decrypt(c) for c in cs))``
(test/test_distributed_keygen.py:132-158) reaches ``PaillierSharedKey.partial_decrypt`` synchronously at
distributed_keygen.py:345-349 and ``.decrypt`` at :378-380, once per coroutine.  On this engine a launch of 1 and a
launch of ~1000 ciphertexts cost the same 13 ms (one wavefront's dependent chain), so m coroutines
that each launch on their own serialise at one-CPU-core speed while the GPU idles.

``Coalescer`` turns the two call sites into awaits.  A coroutine hands over its operand and suspends; the first
submission of a burst schedules a flush with ``loop.call_soon``, i.e. BEHIND every coroutine that is already
runnable, and the flush defers itself while submissions keep arriving (bounded: ``max_defer`` loop iterations, or
``linger`` seconds for pools whose messages trickle in over a network).  The flush then runs every pending
operation of a kind as one batch per key — the batches of different keys (the parties of a ``distributed=False``
run share one process and one GPU, test/conftest.py:73-90) side by side on separate streams — and resolves each
coroutine's future with ITS result or ITS exception: a ciphertext of the wrong type / key raises in the submitting
coroutine before anything is queued (paillier_shared_key.py:62-68), a missing share raises ``KeyError`` there
(:108-110), and a recombination that is not 1 modulo N raises ``ValueError`` only in the coroutine that owns it
(:119-123).

All state belongs to one event loop and one thread (the reference's threading model); the GPU work
itself stays synchronous inside the flush, exactly as the reference's arithmetic blocks its loop.
"""

from __future__ import annotations

import asyncio
import time
import weakref
from typing import Any, Callable, Dict, List, Tuple

NOT_DIVISIBLE = (
    "Combined decryption minus one is not divisible by N. This might be caused by the "
    "fact that the ciphertext that is being decrypted, differs between the parties."
)


class _LoopState:
    __slots__ = ("pending", "count", "seen", "defers", "scheduled")

    def __init__(self) -> None:
        self.pending: Dict[Tuple[str, int], Tuple[Any, List[Tuple[Any, "asyncio.Future"]]]] = {}
        self.count = 0          # submissions since the last flush
        self.seen = 0           # ... as of the previous tick
        self.defers = 0
        self.scheduled = False


class _TickBatcher:
    """Queueing and flush timing shared by the batchers below; subclasses implement ``_execute(pending)``."""

    def __init__(self, engine: Any = None, max_defer: int = 8, linger: float = 0.0) -> None:
        self._engine = engine
        self.max_defer = int(max_defer)
        self.linger = float(linger)
        self._loops: "weakref.WeakKeyDictionary[Any, _LoopState]" = weakref.WeakKeyDictionary()
        self.stats: Dict[str, Any] = {"submitted": 0, "flushes": 0, "busy_s": 0.0}     # busy_s: wall time inside the flushes

    # ------------------------------------------------------------------ queueing
    def _submit(self, kind: str, key: Any, item: Any) -> "asyncio.Future":
        loop = asyncio.get_running_loop()
        state = self._loops.get(loop)
        if state is None:
            state = self._loops[loop] = _LoopState()
        fut = loop.create_future()
        state.pending.setdefault((kind, id(key)), (key, []))[1].append((item, fut))
        state.count += 1
        self.stats["submitted"] += 1
        if not state.scheduled:
            state.scheduled = True
            state.seen = state.defers = 0
            if self.linger > 0:
                loop.call_later(self.linger, self._flush, state)
            else:
                loop.call_soon(self._tick, loop, state)
        return fut

    def _tick(self, loop: Any, state: _LoopState) -> None:
        # still growing: coroutines woken by the same burst of messages are still reaching their call site
        if state.count != state.seen and state.defers < self.max_defer:
            state.seen = state.count
            state.defers += 1
            loop.call_soon(self._tick, loop, state)
            return
        self._flush(state)

    def _flush(self, state: _LoopState) -> None:
        pending, state.pending = state.pending, {}
        state.count = state.seen = state.defers = 0
        state.scheduled = False
        if not pending:
            return
        self.stats["flushes"] += 1
        t0 = time.perf_counter()
        try:
            self._execute(pending)
        finally:
            self.stats["busy_s"] = self.stats.get("busy_s", 0.0) + (time.perf_counter() - t0)

    def _execute(self, pending) -> None:  # pragma: no cover - abstract
        raise NotImplementedError

    def _run(self, groups, executor: Callable, counter: str) -> None:
        """executor(groups) -> one list of results (or exception objects) per group; resolves the futures."""
        try:
            results = executor(groups)
        except BaseException as exc:          # an engine failure belongs to every coroutine of the batch
            for _, entries in groups:
                for _, fut in entries:
                    if not fut.done():
                        fut.set_exception(exc)
            if not isinstance(exc, Exception):
                raise
            return
        for (_, entries), outs in zip(groups, results):
            self.stats[counter] = self.stats.get(counter, 0) + 1
            self.stats["largest_batch"] = max(self.stats.get("largest_batch", 0), len(entries))
            for (_, fut), out in zip(entries, outs):
                if fut.done():                # the coroutine was cancelled while it waited
                    continue
                if isinstance(out, Exception):
                    fut.set_exception(out)
                else:
                    fut.set_result(out)


class Coalescer(_TickBatcher):
    """One per ``patch.install()`` (i.e. per process and engine).  ``stats`` counts what was launched."""

    def __init__(self, engine: Any = None, max_defer: int = 8, linger: float = 0.0) -> None:
        super().__init__(engine, max_defer, linger)
        self.stats.update({"partial_launches": 0, "combine_launches": 0, "largest_batch": 0})

    # ------------------------------------------------------------------ the two awaitable call sites
    async def partial_decrypt(self, key: Any, ciphertext: Any) -> int:
        """``key.partial_decrypt(ciphertext)`` (PSK:52-93) — batched with every other pending one of `key`.
        The checks of PSK:62-68 and the ``get_value()`` side effect (PSK:69) happen here, in the caller."""
        key._check_ciphertext(ciphertext)
        return await self._submit("partial", key, ciphertext.get_value())

    async def decrypt(self, key: Any, partial_dict: Dict[int, int]) -> int:
        """``key.decrypt(partial_dict)`` (PSK:95-127) — batched likewise; ``KeyError`` (PSK:108-110) raised here."""
        row = [partial_dict[i + 1] for i in range(key.share.degree + 1)]
        return await self._submit("combine", key, row)

    # ------------------------------------------------------------------ execution
    def _execute(self, pending) -> None:
        partial = [(key, entries) for (kind, _), (key, entries) in pending.items() if kind == "partial"]
        combine = [(key, entries) for (kind, _), (key, entries) in pending.items() if kind == "combine"]
        if partial:
            self._run(partial, self._partial_groups, "partial_launches")
        if combine:
            self._run(combine, self._combine_groups, "combine_launches")

    @staticmethod
    def _invert_own(key: Any, values: List[int]) -> Tuple[List[int], Dict[int, Exception]]:
        """PSK:89-91 for one key's pending ciphertexts: (inverses of the invertible ones in order, {position: the
        exception of a ciphertext that has no inverse modulo N^2}).  The reference fails exactly the decrypt() that
        holds such a ciphertext; one device inversion of the whole batch fails as a whole (Engine.modinv_batch raises
        for the batch), so on failure the offenders are found on the host (a gcd each, only on this path) and the rest
        is inverted again without them."""
        import math

        try:
            return key.engine.modinv_batch(values, key.n_square), {}
        except ValueError as exc:
            n2 = key.n_square
            bad = {k: ValueError(*exc.args) for k, v in enumerate(values) if math.gcd(v % n2, n2) != 1}
            if not bad:          # not an invertibility failure after all: it belongs to every ciphertext of this key
                raise
            rest = [v for k, v in enumerate(values) if k not in bad]
            return (key.engine.modinv_batch(rest, n2) if rest else []), bad

    @staticmethod
    def _partial_groups(groups) -> List[List[Any]]:
        """One modexp batch per key; several keys side by side when the engine can (Engine.powmod_nsquare_groups).
        A failure stays with what caused it: a ciphertext without an inverse fails its own coroutine, a
        key whose batch the engine refuses fails that key's coroutines — never the burst."""
        results: List[Any] = [None] * len(groups)
        jobs, owners, holes = [], [], []
        for g, (key, entries) in enumerate(groups):
            values = [v for v, _ in entries]
            try:
                exp = key.lagrange_exponent()
                bad: Dict[int, Exception] = {}
                if exp < 0:       # PSK:89-91
                    values, bad = Coalescer._invert_own(key, values)
                    exp = -exp
            except Exception as exc:
                results[g] = [exc] * len(entries)
                continue
            jobs.append((values, exp, key.n))
            owners.append(g)
            holes.append(bad)

        def scatter(g: int, outs: List[Any], bad: Dict[int, Exception]) -> List[Any]:
            it = iter(outs)
            return [bad[k] if k in bad else next(it) for k in range(len(groups[g][1]))]

        live = [j for j, (values, _, _) in enumerate(jobs) if values]
        for j in range(len(jobs)):
            if j not in live:
                results[owners[j]] = scatter(owners[j], [], holes[j])
        if live:
            engine = groups[owners[live[0]]][0].engine
            side_by_side = getattr(engine, "powmod_nsquare_groups", None)
            outs = None
            if side_by_side is not None and len(live) > 1 and all(groups[owners[j]][0].engine is engine for j in live):
                try:
                    outs = side_by_side([jobs[j] for j in live])
                except Exception:
                    outs = None          # one key's batch was refused: run them one by one, the failure stays with its key
            for pos, j in enumerate(live):
                g = owners[j]
                key = groups[g][0]
                try:
                    own = outs[pos] if outs is not None else key.engine.powmod_nsquare_batch(*jobs[j])
                    results[g] = scatter(g, own, holes[j])
                except Exception as exc:
                    results[g] = [exc] * len(groups[g][1])
        return results

    @staticmethod
    def _combine_groups(groups) -> List[List[Any]]:
        out = []
        for key, entries in groups:
            messages, ok = key.engine.combine_batch([row for row, _ in entries], key.n, key.theta_inv)
            out.append([m if good else ValueError(NOT_DIVISIBLE) for m, good in zip(messages, ok)])
        return out


def _same(a: Any, b: Any) -> bool:
    """Equal operands of two co-located parties: the in-process pool hands every party the SAME objects, so this is
    mostly an identity test; otherwise an element-wise comparison in C (never a hash of big integers)."""
    return a is b or a == b


class RoundCoalescer(_TickBatcher):
    """The three device steps of a key-generation round (patch.compute_modulus through biprime.BiprimeRound) for
    parties that share ONE process and GPU — the reference's ``distributed=False`` mode (README.md:83,
    test/conftest.py:73-90), where the n parties' ``compute_modulus`` coroutines interleave on one event loop.  Every
    party reconstructs the same candidate moduli from the same share table, tests the same generators against them
    and votes on the same v values; only its exponents — ``(N - p_1 - q_1 + 1) // 4`` or ``(p_i + q_i) // 4``,
    distributed_keygen.py:1094,1097 — are its own.  So, of the requests pending in the same turn of the loop:

      reconstruct_and_sieve   equal share tables run ONCE; the other parties adopt the result (moduli stay on the device)
      v_calculation           parties with the same survivors and generators share ONE launch: their candidate groups are
                              concatenated (mx_powmod_multi_dev takes a modulus and an exponent per group)
      verdicts                equal v tables run ONCE

    — one launch per kernel and round instead of n.  A party alone in its process (``distributed=True``) gets its
    own launch two loop turns later; results are identical either way."""

    MERGE_MODEXPS = 1 << 16       # beyond this many modexps in total the parties' launches fill the machine on their own

    def __init__(self, engine: Any = None, max_defer: int = 8, linger: float = 0.0, merge: bool = True) -> None:
        super().__init__(engine, max_defer, linger)
        self.merge = bool(merge)      # False: every request runs on its own (A/B runs of the coalescing itself)
        self.stats.update({"sieve_launches": 0, "sieve_requests": 0, "v_launches": 0, "v_requests": 0,
                           "verdict_launches": 0, "verdict_requests": 0})

    async def reconstruct_and_sieve(self, rnd: Any, shares_by_party, prime: int, degree: int, prime_list, points=None):
        return await self._submit("sieve", self, (rnd, shares_by_party, prime, degree, prime_list, points))

    async def v_calculation(self, rnd: Any, g_values, index: int, p_shares, q_shares, keep: int):
        return await self._submit("v", self, (rnd, g_values, index, p_shares, q_shares, keep))

    async def verdicts(self, rnd: Any, v_by_party, keep: int, errors: str = "raise"):
        return await self._submit("verdict", self, (rnd, v_by_party, keep, errors))

    # ------------------------------------------------------------------ execution
    def _execute(self, pending) -> None:
        for kind, fn in (("sieve", self._do_sieve), ("v", self._do_v), ("verdict", self._do_verdict)):
            entries = [e for (k, _), (_, es) in pending.items() if k == kind for e in es]
            if entries:
                self._resolve(entries, fn)

    def _resolve(self, entries, fn) -> None:
        try:
            outs = fn([item for item, _ in entries])
        except BaseException as exc:
            for _, fut in entries:
                if not fut.done():
                    fut.set_exception(exc)
            if not isinstance(exc, Exception):
                raise
            return
        for (_, fut), out in zip(entries, outs):
            if fut.done():
                continue
            if isinstance(out, Exception):
                fut.set_exception(out)
            else:
                fut.set_result(out)

    def _classes(self, items, same) -> List[List[int]]:
        """indices of `items` grouped into classes of mutually `same` requests, in order of first appearance"""
        if not self.merge:
            return [[k] for k in range(len(items))]
        classes: List[List[int]] = []
        for k, it in enumerate(items):
            for cls in classes:
                if same(items[cls[0]], it):
                    cls.append(k)
                    break
            else:
                classes.append([k])
        return classes

    def _do_sieve(self, items) -> List[Any]:
        def same(a, b):
            # the interpolation points as a SET: a party takes the first degree+1 entries of its share dictionary in ITS
            # insertion order (own share first, then the others' in the order they arrived) — Lagrange interpolation over
            # the same set of points of the same table is the same exact sum whatever the order
            pa, pb = a[5], b[5]
            pts = (pa is None and pb is None) or (pa is not None and pb is not None and set(pa) == set(pb))
            return a[2] == b[2] and a[3] == b[3] and pts and _same(a[4], b[4]) and _same(a[1], b[1])

        outs: List[Any] = [None] * len(items)
        self.stats["sieve_requests"] += len(items)
        for cls in self._classes(items, same):
            rnd, shares, prime, degree, prime_list, points = items[cls[0]]
            try:
                first = rnd.reconstruct_and_sieve(shares, prime, degree, prime_list, points=points)
                self.stats["sieve_launches"] += 1
            except Exception as exc:
                for k in cls:
                    outs[k] = exc
                continue
            outs[cls[0]] = first
            for k in cls[1:]:
                outs[k] = items[k][0].adopt_sieve(rnd)
        return outs

    def _do_v(self, items) -> List[Any]:
        from .biprime import BiprimeRound

        def same(a, b):
            return a[5] == b[5] and a[0].shares_survivors_with(b[0]) and _same(a[1], b[1])

        outs: List[Any] = [None] * len(items)
        self.stats["v_requests"] += len(items)
        for cls in self._classes(items, same):
            total = sum(len(items[k][0].moduli) for k in cls) * max(1, items[cls[0]][5])
            try:
                if len(cls) > 1 and total <= self.MERGE_MODEXPS:
                    merged = BiprimeRound.v_calculation_merged([items[k][0] for k in cls], [items[k][1:] for k in cls])
                    self.stats["v_launches"] += 1
                    for k, lists in zip(cls, merged):
                        outs[k] = lists
                else:
                    for k in cls:
                        rnd, g_values, index, p_shares, q_shares, keep = items[k]
                        outs[k] = rnd.v_calculation(g_values, index, p_shares, q_shares, keep)
                        self.stats["v_launches"] += 1
            except Exception as exc:
                for k in cls:
                    outs[k] = exc
        return outs

    def _do_verdict(self, items) -> List[Any]:
        def same(a, b):
            return a[2] == b[2] and a[3] == b[3] and a[0].shares_survivors_with(b[0]) and _same(a[1], b[1])

        outs: List[Any] = [None] * len(items)
        self.stats["verdict_requests"] += len(items)
        for cls in self._classes(items, same):
            rnd, v_by_party, keep, errors = items[cls[0]]
            try:
                res = rnd.verdicts(v_by_party, keep, errors=errors)
                self.stats["verdict_launches"] += 1
            except Exception as exc:
                for k in cls:
                    outs[k] = exc
                continue
            for k in cls:
                outs[k] = res if k == cls[0] else list(res)
        return outs
