# first-layer kernel: weight staging with every load in flight (was: one dependent round trip per weight and thread); soft-argmax
# kernels: coordinate grids requested before the maps.  Tests, then same-box A/B against the previous conv_stem.hip / decoder.hip
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -k "stem or first_layer or u8 or softargmax or sam or context" 2>&1 | tail -2
python -m pytest tests/test_gpu_models.py -q -k "golden or uint8 or u8" 2>&1 | tail -2
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --no-extra-legs --no-clip-leg --no-bf16x3 --steps $3 --warmup 10 --dump-steps gpurun_out/ab_steps.json 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); st=json.load(open('gpurun_out/ab_steps.json')); p=[round(1e3*s['ms'],1) for s in st if 'conv_stem' in (s['kernel'] or '')]; q=[1e3*s['ms'] for s in st if s['kind'] in ('sam','sam_ctx')]
print('$1 $2', d['value'], d['ms_per_step'], 'first layer us', p, 'sam us', round(sum(q)))"
}
V=DEEPHAR_HIP_LIB=$PWD/deephar_amd/csrc/build/variant_stem_old.so
for rep in 1 2 3; do one $V speed2d 200; one X=1 speed2d 200; done
for wl in ntu_spnet mpii penn_merge h36m; do one $V $wl 30; one X=1 $wl 30; done
one $V ntu_spnet 30; one X=1 ntu_spnet 30
