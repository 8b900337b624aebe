for mode in "" "--no-graph"; do for st in 1 2; do
python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 --streams $st --stream-policy tail $mode 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams=$st $mode', d['value'], d['ms_per_step'], 'host enqueue us/step', d['per_rank'][0]['host_enqueue_us_per_step'])"
done; done
