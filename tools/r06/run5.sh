mkdir -p gpurun_out/r5
for g in 0 1; do
DEEPHAR_GROUP_LAUNCHES=$g python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 50 --warmup 10 --streams 1 --dump-steps gpurun_out/r5/steps_speed2d_g$g.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('1 stream group=$g', d['value'], d['ms_per_step'])"
done
