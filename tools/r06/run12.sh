python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "pose_times or resample or siblings or merged_kxk or grouped or spnet_multitask" 2>&1 | tail -3
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "softargmax" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_speed2d.py tests/test_gpu_plan_api.py -x -q -m gpu 2>&1 | tail -2
for k in 0 1 0 1; do
DEEPHAR_FOLD_POSE_MUL=$k python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fold_mul=$k', d['value'], d['ms_per_step'], d['roofline'].get('whole_forward_frac'))"
done
