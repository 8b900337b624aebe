for w in 0 1 0 1; do
DEEPHAR_MERGE_WIDE=$w python bench.py --workload ntu_spnet --no-cpu-baseline --no-predict --steps 20 --warmup 5 --dump-steps gpurun_out/steps_ntu_wide$w.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge_wide=$w', d['value'], d['ms_per_step'], d['roofline'].get('whole_forward_frac'))"
done
python - <<'PY'
import json
a=json.load(open('gpurun_out/steps_ntu_wide0.json')); b=json.load(open('gpurun_out/steps_ntu_wide1.json'))
for s in a:
    if s['name'] in ('res1_shortcut_conv','res1_conv1','res1_conv2'): print('apart', s['name'], round(s['ms']*1e3,1), s['kernel'])
for s in b:
    if 'res1_' in s['name']: print('merged', s['name'], round(s['ms']*1e3,1), s['kernel'])
PY
