# paired skinny convs (dh_conv2d_pair_f32): tests, then same-box A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "conv_pair_launch or grouped_launch" 2>&1 | tail -15
python -m pytest tests/test_gpu_models.py -x -q -k "paired_skinny or grouped_launches" 2>&1 | tail -15
python -m pytest tests/test_gpu_plan_api.py tests/test_gpu_speed2d.py -x -q 2>&1 | tail -8
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --no-extra-legs --steps $3 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 $2', d['value'], d['ms_per_step'])"
}
for rep in 1 2 3; do
one DEEPHAR_PAIR_CONVS=0 speed2d 200
one DEEPHAR_PAIR_CONVS=1 speed2d 200
done
one DEEPHAR_PAIR_CONVS=0 penn_merge 30
one DEEPHAR_PAIR_CONVS=1 penn_merge 30
