for f in 7.5 15 30; do
DEEPHAR_TAIL_KEEP_BACKBONE=1 DEEPHAR_TAIL_ORDER=backbone DEEPHAR_TAIL_FLOOR_US=$f python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('keep-backbone floor=$f', d['value'], d['ms_per_step'], d['config']['streams'])"
done
