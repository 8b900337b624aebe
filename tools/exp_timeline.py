"""Experiment (round 3): the TIMELINE of a two-stream eager forward of the MPII model, measured with HIP events on the
streams themselves (rocprofv3's kernel trace serialises dispatches, so it cannot show overlap).  For every step: the
time its stream reached it and left it, relative to the start of the forward.  Policy 'lowres' puts every <= 16x16 map
on stream 1, so a block reads: [dw_a, GEMM_a] on stream 0 next to the low-resolution chain on stream 1."""
import ctypes as C
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deephar_amd import _lib
from deephar_amd.engine import schedule

policy = sys.argv[1] if len(sys.argv) > 1 else 'lowres'
orig = schedule.assign_streams
def assign(plan, deps, nstreams):
    if policy != 'lowres' or nstreams <= 1:
        return orig(plan, deps, nstreams)
    stream = [0] * len(plan.steps)
    for j, s in enumerate(plan.steps):
        v = next(iter(s.outs.values()), None) if s.outs else None
        if s.kind in ('conv', 'dwconv', 'pool') and v is not None and len(v.shape) >= 3 and \
                v.shape[-3] * v.shape[-2] <= 256 and not s.attrs.get('up2'):
            stream[j] = 1
    return stream
schedule.assign_streams = assign
import bench
m = bench.build_mpii(8); m.num_streams = 2
ex = m.executor; ex.use_graph = False
bp = ex.bind(64)
x = np.random.default_rng(0).uniform(-1, 1, (64, 256, 256, 3)).astype(np.float32)
with torch.cuda.stream(ex.stream):
    ex.set_inputs(bp, [x])
ex.stream.synchronize()
lib = bp.lib
main = ex.stream_ptr
for _ in range(3): bp.launch_all(main)
torch.cuda.synchronize()
side = bp._side_streams()
ptrs = [main] + list(side)
def ev():
    e = C.c_void_p(); _lib.check(lib.dh_event_create(C.byref(e))); return e
n = len(bp.calls)
E0 = ev(); starts = [ev() for _ in range(n)]; ends = [ev() for _ in range(n)]
# replica of BoundPlan.launch_all with timing events around every step on its own stream
_lib.check(lib.dh_event_record(E0, main))
_lib.check(lib.dh_event_record(bp._fork, main))
for st in side: _lib.check(lib.dh_stream_wait_event(st, bp._fork))
for i, (fn, args, step) in enumerate(bp.calls):
    sp = ptrs[step.stream]
    for w in step.wait:
        e = bp._events[w + bp.npre]
        _lib.check(lib.dh_stream_wait_event(sp, e))
    _lib.check(lib.dh_event_record(starts[i], sp))
    _lib.check(fn(*args, sp))
    _lib.check(lib.dh_event_record(ends[i], sp))
    if step.record: _lib.check(lib.dh_event_record(bp._events[i], sp))
for st, e in zip(side, bp._join):
    _lib.check(lib.dh_event_record(e, st)); _lib.check(lib.dh_stream_wait_event(main, e))
torch.cuda.synchronize()
def t(e):
    ms = C.c_float(); _lib.check(lib.dh_event_elapsed_ms(E0, e, C.byref(ms))); return ms.value * 1e3
rows = []
for i, (fn, args, step) in enumerate(bp.calls):
    v = next(iter(step.outs.values()), None) if step.outs else None
    rows.append((t(starts[i]), t(ends[i]), step.stream, step.kind, tuple(v.shape[-3:]) if v is not None else ()))
print('policy %s: forward %.1f us, sum of step times %.1f us' % (policy, max(r[1] for r in rows), sum(r[1] - r[0] for r in rows)))
for s0, e0, st, kind, shp in rows:
    if 3000 < s0 < 5200:
        print('%8.1f %8.1f %7.1f  s%d  %-8s %s' % (s0, e0, e0 - s0, st, kind, shp))
