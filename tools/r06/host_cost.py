"""Host cost vs device time of ONE forward of the speed2d model: launch with the GPU idle, time the host call and the wait."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from deephar_amd import Model
for streams, graph in ((1, True), (2, True), (1, False), (2, False)):
    full = bench.build_speed2d()
    m = Model(full.input, full.outputs[34:36])
    m.num_streams, m.stream_policy = streams, 'tail'
    x = np.random.default_rng(0).uniform(-1, 1, (2, 8, 256, 256, 3)).astype(np.float32)
    m.executor.use_graph = graph
    m.predict(x, batch_size=2)
    ex = m.executor; bp = ex.bound[2]
    host, total = [], []
    for it in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(ex.stream):
            ex.forward(bp)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    print('streams %d graph %s: host call %.0f us (min %.0f), launch-to-done %.0f us (min %.0f), nodes %d' % (
        streams, graph, 1e6 * np.median(host), 1e6 * min(host), 1e6 * np.median(total), 1e6 * min(total), len(bp.calls)))
