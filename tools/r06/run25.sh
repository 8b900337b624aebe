# R14 (pooled segment read by the skinny conv, dh_conv2d_seg_f32): tests, then same-box A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "segments or conv_pair or skinny" 2>&1 | tail -12
python -m pytest tests/test_gpu_models.py -x -q -k "segmented or sibling_pools or paired_skinny or grouped" 2>&1 | tail -12
python -m pytest tests/test_gpu_plan_api.py tests/test_gpu_speed2d.py -x -q 2>&1 | tail -6
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --no-extra-legs --steps $3 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 $2', d['value'], d['ms_per_step'], d['roofline'].get('launches_per_forward'))"
}
for rep in 1 2 3; do
one DEEPHAR_POOL_SEGMENTS=0 speed2d 200
one DEEPHAR_POOL_SEGMENTS=1 speed2d 200
done
one DEEPHAR_POOL_SEGMENTS=0 penn_merge 30
one DEEPHAR_POOL_SEGMENTS=1 penn_merge 30
one DEEPHAR_POOL_SEGMENTS=0 ntu_spnet 20
one DEEPHAR_POOL_SEGMENTS=1 ntu_spnet 20
