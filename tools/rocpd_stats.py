#!/usr/bin/env python
"""Summaries of rocprofv3's rocpd SQLite output (what `--stats` would print as CSV).

    python tools/rocpd_stats.py kernels <out_results.db> [out.csv]     per-kernel calls / total / avg / min / max / %
    python tools/rocpd_stats.py pmc <out_results.db> [substring]      counter sums per kernel launch
"""
import csv
import sqlite3
import sys


def kernels(db, out=None):
    con = sqlite3.connect(db)
    rows = con.execute('''select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start),
                                 min(d.end - d.start), max(d.end - d.start)
                          from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
                          group by s.kernel_name order by 3 desc''').fetchall()
    total = sum(r[2] for r in rows) or 1
    w = csv.writer(open(out, 'w', newline='') if out else sys.stdout)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'MinNs', 'MaxNs', 'Percentage'])
    for name, n, tot, avg, mn, mx in rows:
        w.writerow([name, n, int(tot), round(avg, 1), int(mn), int(mx), round(100.0 * tot / total, 3)])


def pmc(db, sub=''):
    con = sqlite3.connect(db)
    q = '''select s.kernel_name, i.name, sum(p.value), count(distinct p.event_id)
           from rocpd_pmc_event p join rocpd_info_pmc i on p.pmc_id = i.id
           join rocpd_kernel_dispatch d on p.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, i.name'''
    for k, n, v, c in con.execute(q):
        if sub in k:
            print('%-28s %18.0f per launch (%d launches)  %s' % (n, v / c, c, k[:90]))


if __name__ == '__main__':
    {'kernels': kernels, 'pmc': pmc}[sys.argv[1]](*sys.argv[2:])
