python -m pytest tests/test_gpu_plan_api.py -x -q -m gpu 2>&1 | tail -2
for st in 2 3 2 3; do
python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 --streams $st --stream-policy tail 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$st', d['value'], d['ms_per_step'], d['config']['streams'])"
done
python bench.py --workload speed2d --no-cpu-baseline --steps 20 --warmup 5 --speed2d-blocks 7,8,9,13,17 --speed2d-clips 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['speed2d']['blocks'], d['speed2d']['fps_per_block'], d['speed2d']['launches_per_call'])"
