#!/bin/bash
# A/B builds of the library on a long bench run while sampling clock and socket power (rocm-smi):
#   tools/ab_power.sh <steps> <lib A or "tree"> <lib B or "tree"> ...
# One line per run: frames/s, ms per step, dominant-kernel fraction, mean us of the 32x32x576 depthwise / pointwise
# launches (eager pass), median sclk MHz and W over the run.
steps=$1; shift
for lib in "$@"; do
  if [ "$lib" = tree ]; then unset DEEPHAR_HIP_LIB; else export DEEPHAR_HIP_LIB=$PWD/$lib; fi
  tag=$(basename $lib .so)
  ( for i in $(seq 1 60); do rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -n +2 | head -1; sleep 0.2; done ) > gpurun_out/smi_$tag.txt &
  smi=$!
  python bench.py --steps $steps --no-cpu-baseline --no-predict --no-bf16x3 --no-clip-leg --dump-steps gpurun_out/steps_$tag.json 2>/dev/null > gpurun_out/line_$tag.json
  kill $smi 2>/dev/null; wait $smi 2>/dev/null
  python - "$tag" <<'P'
import json, sys, re, statistics as st
tag = sys.argv[1]
d = json.loads(open('gpurun_out/line_%s.json' % tag).read().strip().splitlines()[-1])
steps = json.load(open('gpurun_out/steps_%s.json' % tag))
avg = lambda kind, out: 1e3 * st.mean(s['ms'] for s in steps if s['kind'] == kind and tuple(s['out']) == out)
clk, pw = [], []
for l in open('gpurun_out/smi_%s.txt' % tag):
    m = re.findall(r'\((\d+)Mhz\)', l); w = l.strip().split(',')[-1]
    if len(m) >= 3 and float(w or 0) > 600: clk.append(int(m[2])); pw.append(float(w))
print('%-10s %7.1f fps %7.3f ms  dom %.3f  dw32 %5.1f us  pw32 %6.1f us  sclk %s MHz  %s W (%d samples under load)' % (
    tag, d['value'], d['ms_per_step'], d['roofline']['frac'], avg('dwconv', (32, 32, 576)), avg('conv', (32, 32, 576)),
    int(st.median(clk)) if clk else '-', int(st.median(pw)) if pw else '-', len(clk)))
P
done
