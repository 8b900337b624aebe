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


def assign_streams_tail(plan, deps, items=2, nstreams=2):
    """Streams for the LATENCY regime (a couple of clips per call: exp/pennaction/eval_speed2d.py).  Stream 1 takes a
    SUFFIX A = steps [s, n) of the planner's order: a suffix of a topological order is closed under "is read by", so every
    cross-stream dependency points from stream 0 to stream 1 -- one direction, no ping-pong (the list scheduler above
    produced 367 cross-stream waits on the last block's model of that protocol and ran slower than one stream).  In SPNet
    the suffix is the action stream: nothing in the pose stream reads it (spnet.py:219-248) and the planner emits it behind
    the pose blocks.  Inside A the steps are RE-ORDERED by a list scheduler (earliest ready first): the planner walks the
    action heads from the last prediction block back to the first, but the front end of head k (pose / visual feature
    convolutions, kronecker pooling) only needs pose block k.
    With nstreams >= 3, A is split once more: C = some step x of A and everything in A that (transitively) reads it goes
    to stream 2, the rest F = A - C stays on stream 1 -- again one direction (C reads F and stream 0; F reads stream 0
    only).  For SPNet, C is the chain that carries the action features from head to head (spnet.py:117-131) and F the
    heads' front ends, which then run ahead of the chain.
    s and x minimise the simulated makespan under a per-step cost of a launch floor plus the step's arithmetic / bytes
    for `items` batch items; no split is made when it would not save 5 % of the serial time.
    `items` = 2 and the cost constants (30 us per launch [r06: see below], 90 TFLOP/s, 3 TB/s) are those of the regime the policy exists
    for -- two clips per call on an MI355X.  A Plan serves every batch size it is later bound to, so the split
    is NOT re-tuned per batch: any split is correct and bit-identical (the waits follow the data dependencies), a batch far
    from two clips merely runs a split balanced for two.  Throughput-regime callers keep the default (one stream).
    -> (stream per step, new order of the steps) -- the order stays topological."""
    import heapq
    n = len(plan.steps)
    stream, order = [0] * n, list(range(n))
    if n < 8:
        return stream, order
    # (round 5: ~7.5 us per small launch in a one-stream graph, the entry flow's big convolutions at ~90 TFLOP/s)
    # [r06] re-calibrated against MEASURED two-stream step times: with the round-5 constant of 7.5 us per launch the split left
    # the origin stream idle for the last 1.3 ms of a 5.3 ms forward (tools/r06/stream_timeline.py: the side stream's small
    # launches cost 10-14 us each while another stream shares the chip, and it carries the last pose block behind every older
    # action launch).  Sweep of this constant on the speed2d forward, same box: 7.5 -> 4.13 ms, 10.5 -> 4.05, 15 -> 4.00,
    # 20 ... 40 -> 3.96 ... 3.99, 60 -> 4.03, 100 -> 4.05 (profiles/r06_tail_policy_calibration.txt): the split is mostly a
    # balance of launch COUNTS.  (Also tried there: a critical-path-first order and a backbone-first order in front of the
    # suffix search -- 4.8-5.2 ms and 4.2-4.3 ms: the suffix of the planner's own block-by-block order is the best of the
    # three.)  DEEPHAR_TAIL_FLOOR_US overrides it for such sweeps.
    import os
    floor = float(os.environ.get('DEEPHAR_TAIL_FLOOR_US', '30'))
    cost = [floor + max(st.flops(items) / 90e12, st.bytes(items) / 3e12) * 1e6 for st in plan.steps]
    consumers = [[] for _ in range(n)]
    for j, d in enumerate(deps):
        for i in d:
            consumers[i].append(j)
    fin0 = [0.0] * n                      # finish time of step i if steps [0, i] run back to back on stream 0
    t = 0.0
    for i in range(n):
        t += cost[i]
        fin0[i] = t
    serial = t

    def simulate(s, chain=None, bound=float('inf')):
        """steps [s, n) list-scheduled (earliest ready first) on stream 1 -- and, for the steps in `chain`, on stream 2:
        -> (makespan, {step: (start, stream)})"""
        indeg = [0] * n
        ready_at = [0.0] * n
        heaps = ([], [])
        which = (lambda j: 1 if chain is not None and j in chain else 0)
        for j in range(s, n):
            for i in deps[j]:
                if i >= s:
                    indeg[j] += 1
                else:
                    ready_at[j] = max(ready_at[j], fin0[i])
            if indeg[j] == 0:
                heapq.heappush(heaps[which(j)], (ready_at[j], j))
        clock, placed = [0.0, 0.0], {}
        while heaps[0] or heaps[1]:
            # the queue whose head can start first
            cand = [(max(clock[q], heaps[q][0][0]), q) for q in (0, 1) if heaps[q]]
            _, q = min(cand)
            r, j = heapq.heappop(heaps[q])
            start = max(clock[q], r)
            clock[q] = start + cost[j]
            if clock[q] >= bound:
                return bound, None
            placed[j] = (start, 1 + q)
            for k in consumers[j]:
                ready_at[k] = max(ready_at[k], clock[q])
                indeg[k] -= 1
                if indeg[k] == 0:
                    heapq.heappush(heaps[which(k)], (ready_at[k], k))
        return max(fin0[s - 1], clock[0], clock[1]), placed

    best_s, best, best_placed = 0, serial, None
    spans = {}
    for s in range(1, n):
        span, placed = simulate(s)
        spans[s] = span
        if span < best:
            best_s, best, best_placed = s, span, placed
    if best_s == 0 or best > 0.95 * serial:
        return stream, order
    shift = int(os.environ.get('DEEPHAR_TAIL_SHIFT', '0'))       # A/B aid: move the start of the suffix by this many steps
    if shift:
        best_s = min(max(best_s + shift, 1), n - 1)
        best, best_placed = simulate(best_s)
    if nstreams >= 3:
        # the chain / front-end split is searched jointly with the suffix start: the best two-stream suffix keeps some of
        # the suffix's early steps on stream 0 to balance TWO streams, which is not where three streams balance
        descendants = {}

        def cone(x):
            if x not in descendants:
                chain, stack = {x}, [x]
                while stack:
                    for k in consumers[stack.pop()]:
                        if k not in chain:
                            chain.add(k)
                            stack.append(k)
                descendants[x] = chain
            return descendants[x]
        for s in sorted(spans, key=spans.get)[:24]:
            for x in range(s, n):
                chain = cone(x)
                if len(chain) < 8 or len(chain) > (n - s) - 8:
                    continue
                span, placed = simulate(s, chain, bound=best)
                if placed is not None and span < 0.97 * best:
                    best, best_placed = span, placed
    start = [fin0[i] - cost[i] for i in range(n)]
    for j, (st_, q) in best_placed.items():
        start[j] = st_
        stream[j] = q
    # merged launch order: by simulated start time (a step starts after everything it reads has finished, costs are
    # positive: the order is topological); stream-0 steps keep their relative order
    order = sorted(range(n), key=lambda j: (start[j], j))
    # (Round 6 tried a helper stream for the SIDE branches of the pose stream -- the 21 1x1 shortcut convolutions, which could
    #  run beside the depthwise convolution of their unit: 4.49 -> 6.57 ms.  Each such branch needs a dependency INTO the
    #  origin stream and one back, and a two-way dependency between the queues of a replayed graph costs ~100 us on this
    #  stack; the one-directional suffix above needs a dozen waits in all.  profiles/r06_helper_stream_experiment.txt)
    return [stream[j] for j in order], order


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


def finalize(plan, nstreams=1, policy='list'):
    """Fill step.stream / step.wait (cross-stream dependencies) and place every buffer in the arena.
    policy: 'list' = the list scheduler (throughput regime: branches of an hourglass), 'tail' = assign_streams_tail (two
    streams, latency regime)."""
    deps = compute_deps(plan)
    if policy == 'tail' and nstreams >= 2:
        stream, order = assign_streams_tail(plan, deps, nstreams=nstreams)
        if order != list(range(len(order))):
            plan.steps[:] = [plan.steps[j] for j in order]
            deps = compute_deps(plan)
    else:
        stream = assign_streams(plan, deps, nstreams)
    n = len(plan.steps)
    reach = happens_before(n, deps, stream)
    # streams are in-order queues: waiting for step i of another stream covers every earlier step of that stream, and a
    # wait made by an earlier step of the SAME stream still holds -- only the newest dependency per source stream is an
    # event wait, and only if nothing on this stream waited that far already
    waited = {}
    for j, s in enumerate(plan.steps):
        s.stream = stream[j]
        s.deps = deps[j]
        newest = {}
        for i in deps[j]:
            if stream[i] != stream[j]:
                newest[stream[i]] = max(newest.get(stream[i], -1), i)
        s.wait = []
        for src, i in sorted(newest.items()):
            if waited.get((stream[j], src), -1) < i:
                s.wait.append(i)
                waited[(stream[j], src)] = i
    # steps whose completion somebody on another stream waits for need an event
    needs_event = set(i for s in plan.steps for i in s.wait)
    for j, s in enumerate(plan.steps):
        s.record = j in needs_event
    plan.nstreams = max(stream) + 1 if n else 1
    allocate(plan, reach)
    return plan
