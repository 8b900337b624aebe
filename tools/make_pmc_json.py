#!/usr/bin/env python
"""profiles/pmc_dominant_kernel.json from the PMC summaries tools/profile_round.sh writes (one counter per rocprofv3 pass).
    python tools/make_pmc_json.py gpurun_out/r03_pmc_summary.txt [gpurun_out/r03_pmc_summary_bf16x3.txt] > profiles/pmc_dominant_kernel.json
FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE is doubled (gfx950 reports half of a wide streaming read: MI355X_MICROARCH.md,
HBM section).  bench.py looks the dominant kernel up here by instantiation name and M x K x N."""
import json, re, sys

def demangle(name):
    m = re.search(r'(gemm1x1s?_(?:wide_)?kernel)I(.*?)EEv', name)
    if not m:
        return None
    args = re.findall(r'L([ib])(\d+)E', m.group(2))
    vals = [('true' if v == '1' else 'false') if t == 'b' else v for t, v in args]
    return '%s<%s>' % (m.group(1), ', '.join(vals))

rows = {}
for path in sys.argv[1:]:
    for line in open(path):
        m = re.match(r'(\S+)\s+(\d+) per launch \((\d+) launches\)\s+(\S+)', line)
        if not m:
            continue
        k = demangle(m.group(4))
        if k:
            rows.setdefault(k, {})[m.group(1)] = int(m.group(2))
out = {}
M, K, N = 65536, 576, 576
for k, c in rows.items():
    if not {'FETCH_SIZE', 'WRITE_SIZE'} <= set(c):
        continue
    e = dict(shape_mkn=[M, K, N], fetch_bytes_per_launch=2 * 1024 * c['FETCH_SIZE'], write_bytes_per_launch=1024 * c['WRITE_SIZE'],
             algorithmic_bytes_per_launch=4 * (M * K + K * N + 2 * M * N))
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and 'GRBM_GUI_ACTIVE' in c:
        e['mfma_busy_cycles_per_launch'] = c['SQ_VALU_MFMA_BUSY_CYCLES']
        e['grbm_gui_active_per_launch_all_xcds'] = c['GRBM_GUI_ACTIVE']
        # busy cycles are summed over 1024 SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs
        e['mfma_busy_fraction'] = round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / (c['GRBM_GUI_ACTIVE'] / 8.0), 4)
    e['source'] = ('rocprofv3 --pmc <counter> --kernel-trace, one counter per pass (tools/profile_round.sh: python tools/bench_one.py '
                   '32 576 576 1 1 <cfg> 4 0 [1] = batch 64, 32x32, 576->576 pointwise GEMM, BN + residual epilogue, no ReLU-on-load); '
                   'FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM section); raw: ' + ', '.join(sys.argv[1:]))
    out[k] = e
json.dump(dict(kernels=out), sys.stdout, indent=1)
