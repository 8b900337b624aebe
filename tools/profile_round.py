#!/usr/bin/env python
"""Round profile on the GPU box, every workload of bench.py:

    /usr/local/graft/bin/gpurun --timeout 1500 -- 'python tools/profile_round.py r05 [--workloads mpii,h36m,...] [--skip-pmc]'

Per workload W:
  1. `python bench.py --workload W` (tilings cached in gpurun_out/<tag>_tune_W.json)       -> <tag>_bench_line_W.json
  2. `rocprofv3 --kernel-trace --stats -- python bench.py --workload W` with that cache     -> <tag>_bench_kernel_stats_W.csv
  3. PMC passes (separate runs; FETCH_SIZE and WRITE_SIZE never share a pass: the TCC block has four slots, they cost
     3 + 2) over `bench.py --workload W --replay-step I`, I = roofline.main_shape_step_index of the line of (1): the SAME
     launch -- instantiation, shape, epilogue, pointers -- whose `algorithmic_bytes_per_launch` the line prints.  Only the
     last R dispatches of the kernel (the replays) are summed.
and gpurun_out/<tag>_pmc_dominant_kernel.json with one entry per measured launch (bench.py looks `traffic` up there by
kernel + M x K x N + epilogue).  FETCH_SIZE is KiB and is doubled (gfx950 tallies the 128-byte requests of 16-byte-per-
lane reads at 64 bytes: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is KiB, taken as is.
Copy what is to be judged into profiles/.
"""
import argparse
import json
import os
import sqlite3
import subprocess
import sys
import time

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out')
REPS = 4
LIGHT = ['--no-cpu-baseline', '--no-predict', '--no-clip-leg', '--no-bf16x3']


def sh(cmd, log, cwd=None, timeout=600):
    t0 = time.time()
    with open(log, 'w') as f:
        rc = subprocess.call(cmd, stdout=f, stderr=subprocess.STDOUT, cwd=cwd, timeout=timeout,
                             env=dict(os.environ, TMPDIR='/tmp'))
    print('[%5.0f s] rc=%d %s' % (time.time() - t0, rc, ' '.join(cmd[:12])), flush=True)
    return rc


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        line = line.strip()
        if line.startswith('{'):
            try:
                return json.loads(line)
            except ValueError:
                pass
    return None


def find_db(d):
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith('results.db'):
                return os.path.join(base, f)
    return None


def pmc_last(db, kernel_sub, last):
    """{counter: value per launch} over the last `last` dispatches of the kernels whose name contains kernel_sub."""
    con = sqlite3.connect(db)
    ids = [r[0] for r in con.execute(
        '''select d.event_id from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           where s.kernel_name like ? order by d.start desc limit ?''', ('%' + kernel_sub + '%', last))]
    if not ids:
        return {}, None
    q = '''select i.name, sum(p.value), count(distinct p.event_id) from rocpd_pmc_event p
           join rocpd_info_pmc i on p.pmc_id = i.id where p.event_id in (%s) group by i.name''' % ','.join('?' * len(ids))
    vals = {n: v / c for n, v, c in con.execute(q, ids)}
    name = con.execute('''select s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s
                          on d.kernel_id = s.id where d.event_id = ?''', (ids[0],)).fetchone()[0]
    return vals, name


def pmc_entry(args, w, base, roof, cfg, kname):
    """PMC passes over `bench.py --replay-step` of the workload's main-shape launch (tiling `cfg` forced when given)."""
    head = 'gemm1x1_kernel<4, 1, 1, '
    sub = ('gemm1x1_kernelILi4ELi1ELi1ELi%sE' % kname[len(head)]) if kname.startswith(head) else kname.split('<')[0]
    counters, seen_name = {}, None
    for group in (['FETCH_SIZE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'], ['WRITE_SIZE']):
        d = os.path.join(OUT, '%s_pmc_%s_%s' % (args.tag, w, group[0]))
        sh(['rocprofv3', '--pmc'] + group + ['--kernel-trace', '-d', d, '-o', 'out', '--'] + base + LIGHT +
           ['--replay-step', str(roof['main_shape_step_index']), '--replay-reps', str(REPS)] +
           ([] if cfg is None else ['--replay-cfg', str(cfg)]),
           os.path.join(OUT, '%s_pmc_%s_%s.log' % (args.tag, w, group[0])), cwd='/tmp')
        db = find_db(d)
        if db:
            vals, name = pmc_last(db, sub, REPS)
            counters.update(vals)
            seen_name = name or seen_name
        subprocess.call(['rm', '-rf', d])
    if not {'FETCH_SIZE', 'WRITE_SIZE'} <= set(counters):
        print('PMC incomplete for', w, kname, counters)
        return None
    e = dict(workload=w, kernel=kname, mangled=seen_name, shape_mkn=roof['main_shape_mkn'],
             epilogue=roof['main_shape_epilogue'], step_index=roof['main_shape_step_index'],
             algorithmic_bytes_per_launch=roof['algorithmic_bytes_per_launch'],
             fetch_bytes_per_launch=int(2 * 1024 * counters['FETCH_SIZE']),
             write_bytes_per_launch=int(1024 * counters['WRITE_SIZE']),
             raw_counters_per_launch={k: round(v, 1) for k, v in counters.items()},
             source='rocprofv3 --pmc, FETCH_SIZE(+MFMA busy, GUI active) and WRITE_SIZE in separate passes over `python '
                    'bench.py --workload %s --replay-step %d%s` (tools/profile_round.py): the last %d dispatches = the '
                    'in-model launch replayed; FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE x2 on gfx950 '
                    '(MI355X_MICROARCH.md, HBM section)' % (w + (' ' + args.bench_args if args.bench_args else ''),
                                                            roof['main_shape_step_index'],
                                                            '' if cfg is None else ' --replay-cfg %d' % cfg, REPS))
    e['traffic_over_algorithmic'] = round((e['fetch_bytes_per_launch'] + e['write_bytes_per_launch']) /
                                          e['algorithmic_bytes_per_launch'], 4)
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in counters and counters.get('GRBM_GUI_ACTIVE'):
        # busy cycles are summed over 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
        e['mfma_busy_fraction'] = round(counters['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (counters['GRBM_GUI_ACTIVE'] / 8.0), 4)
    return e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('tag')
    ap.add_argument('--workloads', default='mpii,h36m,penn_merge,ntu_spnet,speed2d')
    ap.add_argument('--skip-pmc', action='store_true')
    ap.add_argument('--skip-stats', action='store_true')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--bench-args', default='', help="extra bench.py arguments for every run, e.g. '--gemm bf16x3'")
    ap.add_argument('--all-tilings', default='mpii,h36m',
                    help='workloads whose dominant GEMM is PMC-profiled on all three <4,1,1,N> tilings (2 passes each)')
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    py, bench = sys.executable, os.path.join(ROOT, 'bench.py')
    entries = []
    for w in args.workloads.split(','):
        tune = os.path.join(OUT, '%s_tune_%s.json' % (args.tag, w))
        if os.path.exists(tune):
            os.remove(tune)
        base = [py, bench, '--workload', w, '--tune-cache', tune] + args.bench_args.split()
        # (1) the line (full default line for mpii: predict boundary, bf16x3 and clip legs, CPU baseline)
        log = os.path.join(OUT, '%s_bench_%s.log' % (args.tag, w))
        sh(base + ['--steps', str(args.steps), '--warmup', '3', '--dump-steps', os.path.join(OUT, '%s_steps_%s.json' % (args.tag, w))],
           log, cwd=ROOT)
        line = last_json_line(log)
        if line is None:
            print('no bench line for', w, '-- see', log)
            continue
        with open(os.path.join(OUT, '%s_bench_line_%s.json' % (args.tag, w)), 'w') as f:
            json.dump(line, f)
        roof = line['roofline']
        print('%s: %.1f frames/s, %.3f ms/step; dominant %s %s %s frac %.3f (whole forward %.3f)' % (
            w, line['value'], line['ms_per_step'], roof['kernel'], roof['main_shape_mkn'], roof['main_shape_epilogue'],
            roof['frac'], roof['whole_forward_frac']), flush=True)
        # (2) kernel stats of the same command
        if not args.skip_stats:
            d = os.path.join(OUT, '%s_prof_%s' % (args.tag, w))
            sh(['rocprofv3', '--kernel-trace', '--stats', '-d', d, '-o', 'out', '--'] + base + LIGHT +
               ['--steps', str(args.steps), '--warmup', '3'], os.path.join(OUT, '%s_prof_%s.log' % (args.tag, w)), cwd='/tmp')
            db = find_db(d)
            if db:
                subprocess.call([py, os.path.join(ROOT, 'tools', 'rocpd_stats.py'), 'kernels', db,
                                 os.path.join(OUT, '%s_bench_kernel_stats_%s.csv' % (args.tag, w))])
            subprocess.call(['rm', '-rf', d])
        # (3) PMC passes over the in-model launch of the main shape
        if args.skip_pmc:
            continue
        # the autotuner alternates between near-equal tilings of the dominant pointwise GEMM from box to box: cover ALL of
        # them [r05: the driver's box picked <4,1,1,1>, which round 4 had not profiled -> `traffic: null`]
        variants = [(None, roof['kernel'])]
        head = 'gemm1x1_kernel<4, 1, 1, '
        if roof['kernel'].startswith(head) and w in args.all_tilings.split(','):
            tn = roof['kernel'][len(head)]
            for other, cfg in (('3', 11), ('2', 12), ('1', 13)):        # bench.TILES index + 9 (the LDS-DMA family)
                if other != tn:
                    variants.append((cfg, roof['kernel'].replace(head + tn, head + other, 1)))
        for cfg, kname in variants:
            e = pmc_entry(args, w, base, roof, cfg, kname)
            if e is not None:
                entries.append(e)
                print(json.dumps(e), flush=True)
    if entries:
        with open(os.path.join(OUT, '%s_pmc_dominant_kernel.json' % args.tag), 'w') as f:
            json.dump(dict(launches=entries), f, indent=1)


if __name__ == '__main__':
    main()
