"""Experiment (round 3): can the hourglass' latency-bound low-resolution chain hide under the big 32x32 branch?
The real MPII model (8 blocks, batch 64) is launched EAGERLY (no hipGraph: a graph replay does not keep stream
priorities or CU masks) on two streams; variants: the engine's default list scheduling, the 'lowres' policy (every
<= 16x16 map on stream 1), a high-priority stream 1, and CU-masked streams (hipExtStreamCreateWithCUMask)."""
import ctypes as C
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = C.CDLL('libamdhip64.so')

def masked_stream(bits=None, priority=0):
    st = C.c_void_p()
    if bits is None:
        rc = hip.hipStreamCreateWithPriority(C.byref(st), 1, priority)      # 1 = hipStreamNonBlocking
    else:
        words = (C.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xffffffff for i in range(8)])
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return st

from deephar_amd.engine import schedule
_default_assign = schedule.assign_streams


def _lowres_assign(plan, deps, nstreams):
    """every <= 16 x 16 map on stream 1, the rest on stream 0 (the experiment's 'lowres' policy)"""
    if nstreams <= 1:
        return _default_assign(plan, deps, nstreams)
    stream = [0] * len(plan.steps)
    for j, s in enumerate(plan.steps):
        v = next(iter(s.outs.values()), None) if s.outs else None
        if s.kind in ('conv', 'dwconv', 'pool') and v is not None and len(v.shape) >= 3 and \
                v.shape[-3] * v.shape[-2] <= 256 and not s.attrs.get('up2'):
            stream[j] = 1
    return stream


def build(policy, streams=2):
    import bench
    schedule.assign_streams = _lowres_assign if policy == 'lowres' else _default_assign
    m = bench.build_mpii(8)
    m.num_streams = streams
    ex = m.executor
    ex.use_graph = False
    bp = ex.bind(64)
    x = np.random.default_rng(0).uniform(-1, 1, (64, 256, 256, 3)).astype(np.float32)
    with torch.cuda.stream(ex.stream):
        ex.set_inputs(bp, [x])
    ex.stream.synchronize()
    return m, ex, bp

def run(ex, bp, main_ptr, reps=10):
    for _ in range(2): bp.launch_all(main_ptr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): bp.launch_all(main_ptr)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

full = (1 << 256) - 1
for policy in (None, 'lowres'):
    m, ex, bp = build(policy)
    n1 = sum(1 for s in m.plan.steps if s.stream == 1)
    print('policy %s: %d of %d steps on stream 1' % (policy, n1, len(m.plan.steps)))
    print('  eager, plain streams            %.3f ms' % run(ex, bp, ex.stream_ptr))
    ex.use_graph = True
    bp.capture(ex.stream_ptr)
    for _ in range(2): bp.replay(ex.stream_ptr)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): bp.replay(ex.stream_ptr)
    torch.cuda.synchronize()
    print('  hipGraph replay                 %.3f ms' % ((time.perf_counter() - t0) / 10 * 1e3))
    ex.use_graph = False
    bp._side_streams()
    bp._streams[0] = masked_stream(None, -1)
    print('  stream 1 high priority          %.3f ms' % run(ex, bp, ex.stream_ptr))
    hi = masked_stream(None, -1)
    bp._streams[0] = masked_stream(None, 0)
    print('  stream 0 high priority          %.3f ms' % run(ex, bp, hi.value))
    for k in (32, 64, 96):
        lo = (1 << k) - 1
        il = sum(1 << i for i in range(0, 256, 256 // k)) if 256 % k == 0 else lo
        for name, mb in (('low bits', lo), ('interleaved', il)):
            bp._streams[0] = masked_stream(mb)
            t_shared = run(ex, bp, ex.stream_ptr)
            main = masked_stream(full ^ mb)
            t_split = run(ex, bp, main.value)
            print('  stream 1 on %3d CUs (%-11s): stream 0 unmasked %.3f ms, stream 0 on the rest %.3f ms' % (
                k, name, t_shared, t_split))
m, ex, bp = build(None, streams=1)
print('one stream eager %.3f ms' % run(ex, bp, ex.stream_ptr))
