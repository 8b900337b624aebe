for f in 20 30 40 60 100 30 40; do
DEEPHAR_TAIL_FLOOR_US=$f python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('floor=$f', d['value'], d['ms_per_step'], d['config']['streams'])"
done
