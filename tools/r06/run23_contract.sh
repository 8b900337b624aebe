# the driver's own commands on a fresh box: default line, and --steps 20 --warmup 5
mkdir -p gpurun_out
( time python bench.py > gpurun_out/r06_contract_default.json 2> gpurun_out/r06_contract_default.err ) 2>&1 | grep real
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_contract_s20w5.json 2> gpurun_out/r06_contract_s20w5.err ) 2>&1 | grep real
python - <<'PY'
import json
for f in ('default','s20w5'):
    txt=open('gpurun_out/r06_contract_%s.json'%f).read().strip().splitlines()
    print(f, 'lines on stdout:', len(txt), 'last is json:', txt[-1].startswith('{'))
    d=json.loads(txt[-1])
    print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','dtype','vs_baseline')})
    print('roofline', {k:d['roofline'].get(k) for k in ('bound','achieved','peak','frac','traffic','whole_forward_frac','launches_per_forward')})
    print('cpu_baseline', d.get('cpu_baseline'))
    for leg in ('h36m','ntu_spnet','speed2d','frame_sharded_clips','bf16x3'):
        v=d.get(leg) or (d.get('legs') or {}).get(leg)
        if isinstance(v,dict): print(leg, {k:v.get(k) for k in ('value','ms_per_step','whole_forward_frac','launches_per_step','fps_per_block')})
PY
