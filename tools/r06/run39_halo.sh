# halo-resident K x K kernel: px / hc of the 12 DMA slots as multiply + shift instead of integer divides -- tests, same-box A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -k "halo" 2>&1 | tail -2
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --no-extra-legs --no-clip-leg --no-bf16x3 --steps $3 --warmup 10 --dump-steps gpurun_out/ab_steps.json 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); st=json.load(open('gpurun_out/ab_steps.json')); p=[round(1e3*s['ms'],1) for s in st if 'halo' in (s['kernel'] or '')]
print('$1 $2', d['value'], d['ms_per_step'], 'halo us', p)"
}
V=DEEPHAR_HIP_LIB=$PWD/deephar_amd/csrc/build/variant_halo_old.so
for rep in 1 2 3; do one $V ntu_spnet 20; one X=1 ntu_spnet 20; done
for rep in 1 2; do one $V speed2d 200; one X=1 speed2d 200; done
