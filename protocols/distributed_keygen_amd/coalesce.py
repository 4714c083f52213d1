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


class Coalescer:
    """One per ``patch.install()`` (i.e. per process and engine).  ``stats`` counts what was launched."""

    def __init__(self, engine: Any = None, max_defer: int = 8, linger: float = 0.0) -> None:
        self._engine = engine
        self.max_defer = int(max_defer)
        self.linger = float(linger)
        self._loops: "weakref.WeakKeyDictionary[Any, _LoopState]" = weakref.WeakKeyDictionary()
        self.stats = {"submitted": 0, "flushes": 0, "partial_launches": 0, "combine_launches": 0, "largest_batch": 0}

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

    # ------------------------------------------------------------------ execution
    def _flush(self, state: _LoopState) -> None:
        pending, state.pending = state.pending, {}
        state.count = state.seen = state.defers = 0
        state.scheduled = False
        if not pending:
            return
        self.stats["flushes"] += 1
        partial = [(key, entries) for (kind, _), (key, entries) in pending.items() if kind == "partial"]
        combine = [(key, entries) for (kind, _), (key, entries) in pending.items() if kind == "combine"]
        if partial:
            self._run(partial, self._partial_groups, "partial_launches")
        if combine:
            self._run(combine, self._combine_groups, "combine_launches")

    def _run(self, groups, executor: Callable, counter: str) -> None:
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
            self.stats[counter] += 1
            self.stats["largest_batch"] = max(self.stats["largest_batch"], len(entries))
            for (_, fut), out in zip(entries, outs):
                if fut.done():                # the coroutine was cancelled while it waited
                    continue
                if isinstance(out, Exception):
                    fut.set_exception(out)
                else:
                    fut.set_result(out)

    @staticmethod
    def _partial_groups(groups) -> List[List[Any]]:
        """One modexp batch per key; several keys side by side when the engine can (Engine.powmod_nsquare_groups)."""
        jobs = []
        for key, entries in groups:
            values = [v for v, _ in entries]
            exp = key.lagrange_exponent()
            if exp < 0:       # PSK:89-91
                values = key.engine.modinv_batch(values, key.n_square)
                exp = -exp
            jobs.append((values, exp, key.n))
        engine = groups[0][0].engine
        side_by_side = getattr(engine, "powmod_nsquare_groups", None)
        if side_by_side is not None and len(jobs) > 1 and all(key.engine is engine for key, _ in groups):
            return side_by_side(jobs)
        return [key.engine.powmod_nsquare_batch(v, e, n) for (key, _), (v, e, n) in zip(groups, jobs)]

    @staticmethod
    def _combine_groups(groups) -> List[List[Any]]:
        out = []
        for key, entries in groups:
            messages, ok = key.engine.combine_batch([row for row, _ in entries], key.n, key.theta_inv)
            out.append([m if good else ValueError(NOT_DIVISIBLE) for m, good in zip(messages, ok)])
        return out
