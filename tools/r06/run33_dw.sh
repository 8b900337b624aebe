# depthwise LDS kernel: every prologue load issued before the first is consumed -- tests, then same-box A/B against the previous dw_lds.h
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -k "dw or depthwise or grouped" 2>&1 | tail -2
python -m pytest tests/test_gpu_speed2d.py -q -k "full_model" 2>&1 | tail -2
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --no-extra-legs --no-clip-leg --no-bf16x3 --steps $3 --warmup 10 --dump-steps gpurun_out/ab_steps.json 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); st=json.load(open('gpurun_out/ab_steps.json')); p=[1e3*s['ms'] for s in st if s['kind']=='dwconv' and s['ms']>0]
print('$1 $2', d['value'], d['ms_per_step'], 'dw launches', len(p), 'sum us', round(sum(p)), 'avg', round(sum(p)/len(p),1))"
}
V=DEEPHAR_HIP_LIB=$PWD/deephar_amd/csrc/build/variant_dw_old.so
for rep in 1 2 3; do one $V speed2d 200; one X=1 speed2d 200; done
for wl in mpii h36m penn_merge ntu_spnet; do one $V $wl 30; one X=1 $wl 30; done
one $V mpii 30; one X=1 mpii 30
