#!/usr/bin/env python
"""Group bench.py --dump-steps output by (kernel family, output resolution, channels): where a forward's time goes.
    python tools/steps_summary.py gpurun_out/steps.json"""
import json, sys, collections
rows = json.load(open(sys.argv[1]))
tot = sum(r['ms'] for r in rows)
g = collections.OrderedDict()
for r in rows:
    o = r['out'] or []
    key = (r['kernel'].split('<')[0] if r['kind'] == 'conv' else r['kind'], tuple(o[-3:]), round(r['gflop'], 1))
    e = g.setdefault(key, [0, 0.0, 0.0, r['kernel']])
    e[0] += 1; e[1] += r['ms']; e[2] += r['gflop']
print('total %.3f ms over %d launches' % (tot, len(rows)))
for k, (n, ms, gf, kn) in sorted(g.items(), key=lambda kv: -kv[1][1])[:28]:
    print('%5.1f %%  %7.3f ms  %3d x %7.1f us  %6.1f TF  %-22s out %-16s %s' % (100 * ms / tot, ms, n, 1e3 * ms / n, gf / ms if ms else 0, k[0], k[1], kn if n == 1 or True else ''))
