python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "siblings or merged_kxk or grouped or heat_map_head or spnet_multitask or multi_stream" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_speed2d.py -x -q -m gpu 2>&1 | tail -2
for k in 0 1 0 1; do
DEEPHAR_MERGE_SIBLINGS=$k python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('merge_siblings=$k', d['value'], d['ms_per_step'], d['roofline'].get('whole_forward_frac'))"
done
