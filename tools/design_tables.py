#!/usr/bin/env python
"""Generate the tables of DESIGN.md that hold MEASURED numbers from the tracked records under profiles/ (the
hand-copied parity table had drifted from profiles/parity_r04.json by 3.5x).

    python tools/design_tables.py            # rewrites the blocks between <!-- BEGIN generated:NAME --> / <!-- END ... -->
    python tools/design_tables.py --check    # exit 1 if DESIGN.md is not what the records say (tests/test_host_logic.py)

Blocks:
  parity      worst |hip - o64|, |o32 - o64|, |hip - o32| per BASELINE configuration, from profiles/parity_r06.json
  workloads   frames/s, ms per step, dominant launch shape + its fraction of the fp32 MFMA peak, whole-forward fraction,
              PMC traffic / algorithmic bytes, CPU stand-in -- from profiles/r06_bench_line_<workload>.json
  speed2d     the reference's own speed protocol (exp/pennaction/eval_speed2d.py): fps per prediction block, the end of the
              previous round beside the end of this one
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')
ROUND = 'r06'
PREV = 'r05'


def _load(name):
    with open(os.path.join(P, name)) as f:
        txt = f.read().strip()
    try:
        return json.loads(txt)
    except ValueError:
        return json.loads(txt.splitlines()[-1])


# (label, predicate on the record's `case` string)
PARITY_ROWS = [
    ('MPII 8 blocks, 2-D context (configs[1]); seeds {0, 1, 2} x 4 frames', lambda c: 'test_reception_mpii_2d_context_parity' in c),
    ('the same against the REFERENCE CODE golden at real size (`rec2d_8`)', lambda c: 'real_size[rec2d_8]' in c),
    ('H36M 3-D, 8 blocks (configs[2]); seeds {0, 1, 2, 31} x 4 frames', lambda c: 'test_cfg3_h36m_8_blocks_vs_oracle' in c),
    ('the same against the REFERENCE CODE golden at real size (`rec3d_8`)', lambda c: 'real_size[rec3d_8]' in c),
    ('Penn merge, T = 16, 4 blocks (configs[3]); seeds {0, 1, 2} x 2 clips', lambda c: 'test_cfg4_penn_merge' in c),
    ('NTU SPNet, T = 32 (configs[4]), fitted heads, 3 seeds x 2 clips, fp32', lambda c: c.startswith('spnet_flat/cfg5_ntu_T32') and c.endswith('/f32')),
    ('the same against the REFERENCE CODE golden at T = 32 / 256 px (`spnet3d_32_s`), fp32', lambda c: c == 'spnet_flat_golden/spnet3d_32_s/f32'),
    ('SPNet shipped Penn config (replica), fitted heads, 3 seeds x 2 clips, fp32', lambda c: c.startswith('spnet_flat/penn_shipped') and c.endswith('/f32')),
    ('the speed protocol\'s model (eval_speed2d.py: 6 pyramids, actions on all six, replica), fitted heads, 3 seeds x 2 clips, fp32',
     lambda c: c.startswith('spnet_flat/speed2d') and c.endswith('/f32')),
    ('the same against the REFERENCE CODE golden at its real size (`spnet2d_speed_s`): full model under 1 / 2 / 3 streams',
     lambda c: c.startswith('speed2d_golden/full/')),
    ('... and every truncated model of the protocol (18 blocks x {1 stream, 2 streams \'tail\'}, two clips per call)',
     lambda c: c.startswith('speed2d_golden/truncated/')),
    ('every SPNet configuration, `bf16x3` mode', lambda c: c.startswith('spnet_flat') and c.endswith('/bf16x3')),
    ('ReceptionNet configurations, `bf16x3` mode (tests/test_gpu_bf16x3.py)', lambda c: 'test_gpu_bf16x3' in c),
]


def parity_block():
    d = _load('parity_%s.json' % ROUND)
    recs = [r for r in d['records'] if r.get('unit') == 'px' and not r.get('stress') and not r.get('sweep')]
    lines = ['| case | records | hip − o64 | o32 − o64 | hip − o32 |', '|---|---|---|---|---|']
    for label, pred in PARITY_ROWS:
        sel = [r for r in recs if pred(r['case'])]
        if not sel:
            continue
        w = [max(r[k] for r in sel) for k in ('hip_vs_o64', 'o32_vs_o64', 'hip_vs_o32')]
        lines.append('| %s | %d | %.1e | %.1e | %.1e |' % (label, len(sel), w[0], w[1], w[2]))
    worst = max(recs, key=lambda r: r['hip_vs_o64'])
    lines.append('')
    lines.append('`profiles/parity_%s.json`: **%d flat px-records, %d above 1e-3 px against fp64, %d above 1e-3 px against the '
                 'fp32 oracle**; worst record %.1e px (`%s`, %s; CPU fp32 oracle on the same vector: %.1e); %d stress records '
                 'reported, %d inside their a-priori tolerance.' % (
                     ROUND, d['flat_px_records'], d['flat_px_records_above_1e3_vs_o64'], d['flat_px_records_above_1e3_vs_o32'],
                     worst['hip_vs_o64'], worst['case'].split('::')[-1], worst['output'], worst['o32_vs_o64'],
                     d['stress_records_reported_not_asserted'], d['stress_records_within_apriori_tolerance']))
    return '\n'.join(lines)


def workloads_block():
    lines = ['| workload | frames/s | ms / step | dominant launch shape (M × K × N, epilogue) → instantiation | its fraction of 157.3 TF '
             '| whole forward | PMC traffic / algorithmic | launches | CPU stand-in (frames/s) |', '|---|---|---|---|---|---|---|---|---|']
    for w in ('mpii', 'h36m', 'penn_merge', 'ntu_spnet', 'speed2d'):
        try:
            d = _load('%s_bench_line_%s.json' % (ROUND, w))
        except OSError:
            continue
        r = d['roofline']
        steps = r.get('launches_per_forward') or len(_load('%s_steps_%s.json' % (ROUND, w)))
        cpu = d.get('cpu_baseline', {}).get('value')
        lines.append('| %s | %.0f | %.2f | %s, %s → `%s` | %.3f | %.3f | %s | %d | %s |' % (
            w, d['value'], d['ms_per_step'], ' × '.join(str(v) for v in r['main_shape_mkn']), r['main_shape_epilogue'], r['kernel'],
            r['frac'], r['whole_forward_frac'], ('%.3f' % r['traffic_over_algorithmic']) if r.get('traffic') else 'n/a', steps,
            ('%.1f' % cpu) if cpu else 'n/a'))
    return '\n'.join(lines)


def speed2d_block():
    first = _load('%s_bench_line_speed2d.json' % PREV)
    last = _load('%s_bench_line_speed2d.json' % ROUND)
    f, l = first['speed2d'], last['speed2d']
    lines = ['| prediction block b (outputs 2b, 2b + 1) | ' + ' | '.join(str(b) for b in l['blocks']) + ' |',
             '|---|' + '---|' * len(l['blocks']),
             '| launches per call, end of round 5 | ' + ' | '.join(str(f['launches_per_call'][f['blocks'].index(b)]) for b in l['blocks']) + ' |',
             '| launches per call, end of round 6 | ' + ' | '.join(str(v) for v in l['launches_per_call']) + ' |',
             '| frames/s, end of round 5 | ' + ' | '.join('%.0f' % f['fps_per_block'][f['blocks'].index(b)] for b in l['blocks']) + ' |',
             '| frames/s, end of round 6 | ' + ' | '.join('%.0f' % v for v in l['fps_per_block']) + ' |',
             '| ratio | ' + ' | '.join('%.2f' % (v / f['fps_per_block'][f['blocks'].index(b)]) for b, v in zip(l['blocks'], l['fps_per_block'])) + ' |',
             '| fraction of the fp32 MFMA peak, end of round 6 | ' + ' | '.join('%.3f' % v for v in l['whole_forward_frac_per_block']) + ' |',
             '',
             'Device-resident step of the last block\'s model (2 clips = 16 frames): **%.2f ms → %.2f ms (× %.2f)**, %.0f → %.0f '
             'frames/s; `predict` on host arrays, last block: %.0f → %.0f frames/s; CPU stand-in on this host: %s frames/s.' % (
                 first['ms_per_step'], last['ms_per_step'], first['ms_per_step'] / last['ms_per_step'], first['value'], last['value'],
                 f['fps_per_block'][-1], l['fps_per_block'][-1], last.get('cpu_baseline', first.get('cpu_baseline', {})).get('value', 'n/a'))]
    return '\n'.join(lines)


BLOCKS = {'parity': parity_block, 'workloads': workloads_block, 'speed2d': speed2d_block}


def render(text):
    for name, fn in BLOCKS.items():
        pat = re.compile(r'(<!-- BEGIN generated:%s -->\n)(.*?)(<!-- END generated:%s -->)' % (name, name), re.S)
        if not pat.search(text):
            continue
        body = fn() + '\n'
        text = pat.sub(lambda m: m.group(1) + body + m.group(3), text)
    return text


def main():
    path = os.path.join(ROOT, 'DESIGN.md')
    with open(path) as f:
        old = f.read()
    new = render(old)
    if '--check' in sys.argv:
        if new != old:
            print('DESIGN.md is out of date with profiles/: run python tools/design_tables.py')
            sys.exit(1)
        print('DESIGN.md tables match profiles/')
        return
    with open(path, 'w') as f:
        f.write(new)
    print('DESIGN.md tables regenerated from profiles/')


if __name__ == '__main__':
    main()
