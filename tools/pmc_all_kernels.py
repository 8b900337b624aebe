#!/usr/bin/env python
"""HBM-side traffic of EVERY kernel of a workload's forward (not only the dominant one): two rocprofv3 --pmc passes (FETCH_SIZE,
WRITE_SIZE -- never in one pass) over an EAGER `bench.py --workload W --no-graph` run, summed per kernel over the last forward,
beside the algorithmic bytes of the same steps (bench.py --dump-steps).  A kernel that moves much more than its operands is the
first thing to fix (the split-bf16 wide tiling's register spills were found this way, profiles/r06_bf16x3_wide_epilogue_spills.md).

    /usr/local/graft/bin/gpurun -- 'python tools/pmc_all_kernels.py mpii h36m'          -> gpurun_out/pmc_all_kernels_<W>.json
Single-plan (frame) workloads only: the dispatches of the last eager forward are matched to the plan's steps in launch order
(`name_mismatches` must be 0); the clip workloads run two plans per step and are not aligned by this tool.
FETCH_SIZE is KiB and doubled (gfx950: 128-byte requests tallied at 64 bytes -- MI355X_MICROARCH.md); WRITE_SIZE KiB as is."""
import json, os, sqlite3, subprocess, sys, collections

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from profile_round import find_db, LIGHT          # noqa: E402


def last_forward(db, counter, nlaunch):
    """[(kernel name, counter value)] of the last `nlaunch` dispatches of the library's kernels, in launch order: the last
    pass of bench.py's per-step profile = one eager forward"""
    con = sqlite3.connect(db)
    rows = con.execute('''select d.event_id, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                          on d.kernel_id = s.id where s.kernel_name like '%_ZN2dh%' order by d.start desc limit ?''', (nlaunch,)).fetchall()
    vals = dict(con.execute('''select p.event_id, sum(p.value) from rocpd_pmc_event p join rocpd_info_pmc i on p.pmc_id = i.id
                               where i.name = ? group by p.event_id''', (counter,)).fetchall())
    return [(name, vals.get(ev, 0.0)) for ev, name in reversed(rows)]


BASE = dict(pool='pool_kernel', dwconv='dwconv', eltwise='eltwise_kernel', upsample_add='upsample2x_add', kronecker='kronecker',
            globalmaxmin='global_maxmin', copy='copy_channels', zeropad='zeropad', sam='softargmax2d', sam_ctx='softargmax2d',
            depthsum='depth_from_maps', depthmean='depth_means', softargmax1d='softargmax1d')


def main():
    os.makedirs(OUT, exist_ok=True)
    for w in sys.argv[1:]:
        steps = os.path.join(OUT, 'pmcall_steps_%s.json' % w)
        tune = os.path.join(OUT, 'pmcall_tune_%s.json' % w)
        base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', w, '--tune-cache', tune] + LIGHT + \
            ['--no-graph', '--steps', '2', '--warmup', '1']
        subprocess.call(base + ['--dump-steps', steps], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
        dump = [s for s in json.load(open(steps)) if not s.get('absorbed')]
        n = len(dump)
        seq = {}
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(OUT, 'pmcall_%s_%s' % (w, counter))
            subprocess.call(['rocprofv3', '--pmc', counter, '--kernel-trace', '-d', d, '-o', 'out', '--'] + base,
                            stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'))
            db = find_db(d)
            seq[counter] = last_forward(db, counter, n) if db else []
            subprocess.call(['rm', '-rf', d])
        if len(seq['FETCH_SIZE']) != n or len(seq['WRITE_SIZE']) != n:
            print(w, 'PMC incomplete', len(seq['FETCH_SIZE']), len(seq['WRITE_SIZE']), n)
            continue
        rows, bad = [], 0
        for s, (kn, f), (kn2, wv) in zip(dump, seq['FETCH_SIZE'], seq['WRITE_SIZE']):
            want = (s['kernel'] or s['kind']).split('<')[0].split(' ')[0]
            want = BASE.get(want, want)
            if want not in kn or kn != kn2:
                bad += 1
            traffic = (2 * 1024 * f + 1024 * wv) / 1e6
            rows.append(dict(step=s['name'], kernel=s['kernel'] or s['kind'], out=s['out'], eager_us=round(1e3 * s['ms'], 1),
                             algorithmic_MB=round(s['mbytes'], 2), fetch_MB=round(2 * 1024 * f / 1e6, 2), write_MB=round(1024 * wv / 1e6, 2),
                             ratio=round(traffic / s['mbytes'], 3) if s['mbytes'] else None, profiler_kernel=kn))
        tot_alg = sum(r['algorithmic_MB'] for r in rows)
        tot = sum(r['fetch_MB'] + r['write_MB'] for r in rows)
        json.dump(dict(workload=w, launches=n, name_mismatches=bad, total_algorithmic_MB=round(tot_alg, 1), total_traffic_MB=round(tot, 1),
                       steps=rows), open(os.path.join(OUT, 'pmc_all_kernels_%s.json' % w), 'w'), indent=1)
        print('==', w, n, 'launches; name mismatches', bad, '; traffic %.0f MB / algorithmic %.0f MB = %.3f' % (tot, tot_alg, tot / tot_alg))
        # the launches that move the most bytes beyond their operands
        for r in sorted(rows, key=lambda r: -(r['fetch_MB'] + r['write_MB'] - r['algorithmic_MB']))[:25]:
            print('  %-44s %-52s alg %8.1f fetch %8.1f write %8.1f ratio %s  %6.1f us' % ((r['step'] or '')[:44], r['kernel'][:52],
                  r['algorithmic_MB'], r['fetch_MB'], r['write_MB'], r['ratio'], r['eager_us']))


if __name__ == '__main__':
    main()
