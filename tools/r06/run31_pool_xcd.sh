# XCD-contiguous block order of the pooling kernel: tests, per-launch PMC, same-box A/B against the previous spatial.hip
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -k "pool" 2>&1 | tail -2
python tools/pmc_all_kernels.py mpii 2>&1 | grep -E "^==| pool " | head -4
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --no-extra-legs --no-clip-leg --no-bf16x3 --steps $3 --warmup 10 --dump-steps gpurun_out/ab_steps.json 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); st=json.load(open('gpurun_out/ab_steps.json')); p=[round(1e3*s['ms'],1) for s in st if s['kind']=='pool']
print('$1 $2', d['value'], d['ms_per_step'], 'pool us', p[:6])"
}
V=DEEPHAR_HIP_LIB=$PWD/deephar_amd/csrc/build/variant_pool_noxcd.so
for rep in 1 2; do one $V ntu_spnet 20; one X=1 ntu_spnet 20; done
one $V mpii 40; one X=1 mpii 40
one $V speed2d 200; one X=1 speed2d 200
