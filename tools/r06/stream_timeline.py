"""Two-stream timeline of the speed2d forward, measured with HIP timing events recorded in front of every launch on the
launch's own stream (eager launches, no profiler: the streams really overlap).  Prints when each stream finishes, the time
each stream spends per step kind, and the gaps where a stream waits for the other."""
import os, sys, json, collections
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ctypes as C
import numpy as np, torch
import bench
from deephar_amd import Model, _lib

streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
full = bench.build_speed2d()
m = Model(full.input, full.outputs[34:36])
m.num_streams, m.stream_policy = streams, 'tail'
x = np.random.default_rng(0).uniform(-1, 1, (2, 8, 256, 256, 3)).astype(np.float32)
m.predict(x, batch_size=2)
ex = m.executor
bp = ex.bound[2]
lib = bp.lib
ex.use_graph = False
nrep = 5
acc = None
for rep in range(nrep + 1):
    evs = []
    def perturb(i, step, sp, ptrs):
        e = C.c_void_p(); _lib.check(lib.dh_event_create(C.byref(e))); _lib.check(lib.dh_event_record(e, sp)); evs.append((i, step, e))
    bp.perturb = perturb
    with torch.cuda.stream(ex.stream):
        bp.launch_all(ex.stream_ptr)
        end = C.c_void_p(); _lib.check(lib.dh_event_create(C.byref(end))); _lib.check(lib.dh_event_record(end, ex.stream_ptr))
    torch.cuda.synchronize()
    if rep == 0:
        continue
    t0 = evs[0][2]
    ms = C.c_float()
    starts = []
    for i, step, e in evs:
        _lib.check(lib.dh_event_elapsed_ms(t0, e, C.byref(ms))); starts.append(ms.value * 1e3)
    _lib.check(lib.dh_event_elapsed_ms(t0, end, C.byref(ms))); total = ms.value * 1e3
    a = np.array(starts + [total])
    acc = a if acc is None else acc + a
acc /= nrep
steps = [s for _, s, _ in evs]
print('launches', len(steps), 'eager 2-stream forward %.0f us' % acc[-1])
by_stream = collections.defaultdict(list)
for k, s in enumerate(steps):
    by_stream[s.stream].append(k)
for st, idx in sorted(by_stream.items()):
    # a step's span on its stream = start of the next step on the same stream - its own start
    spans = [(acc[idx[j + 1]] if j + 1 < len(idx) else acc[-1]) - acc[idx[j]] for j in range(len(idx))]
    kinds = collections.defaultdict(lambda: [0, 0.0])
    for k, sp in zip(idx, spans):
        key = steps[k].kind
        kinds[key][0] += 1; kinds[key][1] += sp
    print('stream', st, 'steps', len(idx), 'first start %.0f last start %.0f us' % (acc[idx[0]], acc[idx[-1]]),
          {k: (v[0], round(v[1])) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1][1])})
    big = sorted(((sp, steps[k].name, steps[k].kind, round(acc[k])) for k, sp in zip(idx, spans)), reverse=True)[:8]
    print('    longest spans:', [(round(a), b, c, d) for a, b, c, d in big])
# where is the pose chain at the time the action stream ends, and vice versa
last0 = max(k for k, s in enumerate(steps) if s.stream == 0)
print('last stream-0 step', steps[last0].name, 'starts at %.0f us; forward ends at %.0f us' % (acc[last0], acc[-1]))
json.dump(dict(starts=acc.tolist(), names=[s.name for s in steps], kinds=[s.kind for s in steps], streams=[s.stream for s in steps]),
          open(os.path.join(ROOT, 'gpurun_out', 'stream_timeline_%d.json' % streams), 'w'))
