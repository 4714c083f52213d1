#!/usr/bin/env python3
"""Discrete-event model of the unit scheduler of a time-sliced pair-kernel launch (csrc/mx_powmod_n2_split.hpp): `chains`
groups of ciphertexts, each a chain of `s` segments that must run in order, on `workers` resident wavefront pairs; a
segment takes T / s (T = the duration of a plain launch whose groups all fit).  Two disciplines:

  fifo   one queue over all units: first segments in group order, every finished segment appends the group's next one
         (rounds 3-4)
  lrf    one queue per level (segments done so far), a free pair takes from the lowest level that has an entry: the
         group with the most work left first (round 5)

usage: ts_schedule_model.py [workers] [T_ms]      prints the modelled duration over chains x s for both."""
import heapq
import random
import sys


def simulate(chains, workers, s, T=32.0, policy="lrf", jitter=0.02, seed=1):
    rng = random.Random(seed)
    unit = lambda: (T / s) * (1 + rng.uniform(-jitter, jitter))
    events, end = [], 0.0
    if policy == "fifo":
        ring, head, waiting = [], 0, {}
        total = chains * s

        def take(w, t):
            nonlocal head
            idx, head = head, head + 1
            if idx >= total:
                return
            if idx < chains:
                heapq.heappush(events, (t + unit(), w, (idx, 0)))
            elif idx - chains < len(ring):
                heapq.heappush(events, (t + unit(), w, ring[idx - chains]))
            else:
                waiting[idx - chains] = w              # waits for THAT entry

        for w in range(workers):
            take(w, 0.0)
        while events:
            t, w, (c, k) = heapq.heappop(events)
            end = t
            if k + 1 < s:
                ring.append((c, k + 1))
                j = len(ring) - 1
                if j in waiting:
                    heapq.heappush(events, (t + unit(), waiting.pop(j), ring[j]))
            take(w, t)
        return end
    levels = [list(range(chains))] + [[] for _ in range(s)]
    heads = [0] * (s + 1)
    idle = list(range(workers))

    def dispatch(t):
        while idle:
            for k in range(s):
                if heads[k] < len(levels[k]):
                    c = levels[k][heads[k]]
                    heads[k] += 1
                    heapq.heappush(events, (t + unit(), idle.pop(), (c, k)))
                    break
            else:
                return

    dispatch(0.0)
    while events:
        t, w, (c, k) = heapq.heappop(events)
        end = t
        levels[k + 1].append(c)
        idle.append(w)
        dispatch(t)
    return end


if __name__ == "__main__":
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    T = float(sys.argv[2]) if len(sys.argv) > 2 else 32.0
    segs = (2, 3, 4, 5, 6, 8, 12, 16)
    print(f"{workers} resident pairs, plain launch of <= {workers} groups: {T} ms; columns: units per group")
    for policy in ("fifo", "lrf"):
        print(policy)
        for chains in (workers * 9 // 8, 625 * workers // 512, workers * 5 // 4, workers * 11 // 8, workers * 3 // 2, workers * 2):
            print(f"  {chains:5d} groups (fluid {T * chains / workers:5.1f}): " + " ".join(f"s{s} {simulate(chains, workers, s, T, policy):5.1f}" for s in segs))
