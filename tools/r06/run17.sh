for st in 3 3; do
python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 --streams $st --stream-policy tail 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$st', d['value'], d['ms_per_step'], d['config']['streams'])"
done
