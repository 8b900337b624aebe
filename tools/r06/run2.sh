mkdir -p gpurun_out/r2
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dwconv" > gpurun_out/r2/ops.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/r2/ops.log
timeout 900 python -m pytest tests/test_gpu_speed2d.py -x -q -m gpu > gpurun_out/r2/speed2d_tests.log 2>&1; echo "speed2d tests rc=$?"; tail -2 gpurun_out/r2/speed2d_tests.log
for up in 0 1; do
  DEEPHAR_UP_COMMUTE=$up python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 100 --warmup 20 --dump-steps gpurun_out/r2/steps_speed2d_up$up.json > gpurun_out/r2/speed2d_up$up.json 2> gpurun_out/r2/speed2d_up$up.err; echo "speed2d up=$up rc=$?"
  DEEPHAR_UP_COMMUTE=$up python bench.py --workload ntu_spnet --no-cpu-baseline --no-predict --steps 20 --warmup 5 --dump-steps gpurun_out/r2/steps_ntu_up$up.json > gpurun_out/r2/ntu_up$up.json 2> gpurun_out/r2/ntu_up$up.err; echo "ntu up=$up rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r2/*_up?.json')):
    if 'steps_' in f: continue
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline'].get('whole_forward_frac'))
    except Exception as e: print(f, 'ERR', e)
PY
