#!/usr/bin/env python
"""Idle time inside replayed forward graphs, from a rocprofv3 --kernel-trace database:
    python tools/graph_gaps.py <results.db> <launches per step>
Takes the last complete steps of the trace (the timed replays), reports span, busy time (union of kernel intervals),
the sum of kernel durations and the gaps between consecutive kernels."""
import sqlite3, sys
import numpy as np
db, per = sys.argv[1], int(sys.argv[2])
con = sqlite3.connect(db)
rows = con.execute('select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s '
                   'on d.kernel_id = s.id order by d.start').fetchall()
rows = rows[-per * 10:]                                   # ten steps from the end
st = np.array([r[0] for r in rows], dtype=np.int64); en = np.array([r[1] for r in rows], dtype=np.int64)
span = en.max() - st.min()
busy, cur_s, cur_e = 0, st[0], en[0]
for s, e in zip(st[1:], en[1:]):
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
gaps = np.maximum(st[1:] - np.maximum.accumulate(en[:-1]), 0)
print('steps 10  span/step %.3f ms  busy/step %.3f ms  sum of kernel durations/step %.3f ms' % (span / 1e7, busy / 1e7, (en - st).sum() / 1e7))
print('idle/step %.3f ms in %d gaps; median gap %.2f us, mean %.2f us, p90 %.2f us' % ((span - busy) / 1e7, (gaps > 0).sum() // 10, np.median(gaps[gaps > 0]) / 1e3, gaps[gaps > 0].mean() / 1e3, np.percentile(gaps[gaps > 0], 90) / 1e3))

import collections
short = lambda n: n.split('(')[0].replace('_ZN2dh12_GLOBAL__N_1', '')[:44]
c = collections.Counter()
for i in np.nonzero(gaps > 0)[0]:
    c[(short(rows[i][2]), short(rows[i + 1][2]))] += 1
for (a, b), n in c.most_common(14):
    print('%4d  %-46s -> %s' % (n // 10, a, b))
nogap = collections.Counter()
for i in np.nonzero(gaps == 0)[0]:
    nogap[(short(rows[i][2]), short(rows[i + 1][2]))] += 1
print('no gap:')
for (a, b), n in nogap.most_common(8):
    print('%4d  %-46s -> %s' % (n // 10, a, b))
