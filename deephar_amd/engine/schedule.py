"""Stream scheduling + activation memory allocation for a lowered plan (host logic, no GPU needed).

The step list the planner emits is a valid sequential program.  Independent branches of the model (the
full-resolution and the low-resolution arms of an hourglass, reception.py:106-127; the soft-argmax decoder next
to the fReMap re-injection, reception.py:286-312; SPNet's action stream next to its pose stream) are put on
different HIP streams so that the small, latency-bound kernels of one branch fill the chip next to the large
GEMMs of the other.  The whole multi-stream launch sequence is still captured into ONE hipGraph (fork/join by
events), so replay cost does not change.

  deps    : true data dependencies between steps, from the buffers (channel ranges) they read and write
  streams : list scheduling -- a step continues the stream whose tail it depends on, unless a later consumer of
            that tail lies on a longer downstream path (then the later one inherits the stream)
  memory  : two buffers may share arena space only if every access of one happens-before every access of the
            other in the partial order (deps + same-stream order); with one stream this degenerates to the
            classic live-interval packing.
"""
from collections import defaultdict

import numpy as np


def _ranges_overlap(a, b):
    return a[0] < b[1] and b[0] < a[1]


def _step_cost_us(step):
    """Rough duration model used only to rank branches (never reported)."""
    return max(step.flops(64) / 100e12, step.bytes(64) / 3e12) * 1e6 + 5.0


def compute_deps(plan):
    """deps[j] = sorted list of earlier steps j must wait for (RAW, WAR, WAW on overlapping channel ranges)."""
    writers = defaultdict(list)   # buf id -> [(range, step)]
    readers = defaultdict(list)
    deps = []
    for j, s in enumerate(plan.steps):
        d = set()
        for v in s.ins.values():
            if v is None:
                continue
            r = (v.coff, v.coff + v.C)
            for (wr, i) in writers[id(v.buf)]:
                if _ranges_overlap(r, wr):
                    d.add(i)
        for v in s.outs.values():
            if v is None:
                continue
            r = (v.coff, v.coff + v.C)
            for (wr, i) in writers[id(v.buf)]:
                if _ranges_overlap(r, wr):
                    d.add(i)
            for (rr, i) in readers[id(v.buf)]:
                if _ranges_overlap(r, rr):
                    d.add(i)
        d.discard(j)
        deps.append(sorted(d))
        for v in s.ins.values():
            if v is not None:
                readers[id(v.buf)].append(((v.coff, v.coff + v.C), j))
        for v in s.outs.values():
            if v is not None:
                writers[id(v.buf)].append(((v.coff, v.coff + v.C), j))
    return deps


def assign_streams(plan, deps, nstreams):
    n = len(plan.steps)
    stream = [0] * n
    if nstreams <= 1 or n == 0:
        return stream
    consumers = defaultdict(list)
    for j, d in enumerate(deps):
        for i in d:
            consumers[i].append(j)
    # downstream critical-path length
    cost = [_step_cost_us(s) for s in plan.steps]
    down = [0.0] * n
    for i in range(n - 1, -1, -1):
        down[i] = cost[i] + max((down[j] for j in consumers[i]), default=0.0)
    tail = [-1] * nstreams        # last step on each stream
    for j in range(n):
        cands = [s for s in range(nstreams) if tail[s] >= 0 and tail[s] in deps[j]]
        chosen = None
        for s in sorted(cands, key=lambda s: -tail[s]):
            t = tail[s]
            # yield the stream to a later, more critical consumer of the same tail
            rival = max((down[k] for k in consumers[t] if k > j and stream_unassigned(k, j)), default=-1.0)
            if rival <= down[j]:
                chosen = s
                break
        if chosen is None:
            if j == 0:
                chosen = 0
            else:
                free = [s for s in range(nstreams) if s not in cands]
                pool = free if free else list(range(nstreams))
                chosen = min(pool, key=lambda s: tail[s])     # least recently used
        stream[j] = chosen
        tail[chosen] = j
    return stream


def stream_unassigned(k, j):
    return k > j


def happens_before(n, deps, stream):
    """reach[i] = bitmask of steps that are ordered after step i (descendants in deps + same-stream order)."""
    succ = [[] for _ in range(n)]
    for j, d in enumerate(deps):
        for i in d:
            succ[i].append(j)
    last = {}
    for j in range(n):
        s = stream[j]
        if s in last:
            succ[last[s]].append(j)
        last[s] = j
    reach = [0] * n
    for i in range(n - 1, -1, -1):
        m = 0
        for j in succ[i]:
            m |= (1 << j) | reach[j]
        reach[i] = m
    return reach


def allocate(plan, reach):
    """Greedy packing (largest first) under the happens-before conflict rule."""
    n = len(plan.steps)
    acc = defaultdict(list)
    for j, s in enumerate(plan.steps):
        for v in list(s.ins.values()) + list(s.outs.values()):
            if v is not None and j not in acc[id(v.buf)]:
                acc[id(v.buf)].append(j)
    full = (1 << n) - 1
    info = {}
    for b in plan.bufs:
        steps = acc.get(id(b), [])
        mask = 0
        for j in steps:
            mask |= 1 << j
        # everything that is NOT ordered after all of this buffer's accesses
        after_all = full
        for j in steps:
            after_all &= reach[j]
        info[id(b)] = (steps, mask, after_all)
        b.start = min(steps) if steps else 0
        b.end = max(steps) if steps else n
        if b.kind == 'input':
            b.start = -1
        if b.pinned or not steps:
            b.end = n

    def ordered(x, y):
        """all accesses of x happen before all accesses of y"""
        if x.pinned or y.kind == 'input':
            return False
        sx, mx, after_x = info[id(x)]
        sy, my, _ = info[id(y)]
        if not sx or not sy:
            return False
        return (my & ~after_x) == 0

    # model outputs first, back to back from offset 0: ONE device-to-host copy of [0, out_items * n) returns them all
    placed = []
    off = 0
    for b in plan.bufs:
        if b.pinned:
            b.offset = off
            off += b.items
            placed.append(b)
    plan.out_items = off
    for b in sorted(plan.bufs, key=lambda b: -b.items):
        if b.pinned:
            continue
        spans = sorted((p.offset, p.offset + p.items) for p in placed if not (ordered(p, b) or ordered(b, p)))
        off = 0
        for lo, hi in spans:
            if off + b.items <= lo:
                break
            off = max(off, hi)
        b.offset = off
        placed.append(b)
    plan.arena_items = max((b.offset + b.items for b in plan.bufs), default=0)


def finalize(plan, nstreams=1):
    """Fill step.stream / step.wait (cross-stream dependencies) and place every buffer in the arena."""
    deps = compute_deps(plan)
    stream = assign_streams(plan, deps, nstreams)
    n = len(plan.steps)
    reach = happens_before(n, deps, stream)
    for j, s in enumerate(plan.steps):
        s.stream = stream[j]
        s.deps = deps[j]
        s.wait = [i for i in deps[j] if stream[i] != stream[j]]
    # steps whose completion somebody on another stream waits for need an event
    needs_event = set(i for s in plan.steps for i in s.wait)
    for j, s in enumerate(plan.steps):
        s.record = j in needs_event
    plan.nstreams = max(stream) + 1 if n else 1
    allocate(plan, reach)
    return plan
