mkdir -p gpurun_out/r3
for hs in 0 1 0 1; do
  DEEPHAR_HELPER_STREAM=$hs python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 2> gpurun_out/r3/err_$hs.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('helper=$hs', d['value'], d['ms_per_step'], d['config']['streams'])"
done
