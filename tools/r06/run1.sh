mkdir -p gpurun_out/r1
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "ksplit or pointwise or upsample or three_way or dma_gemm or half_resolution or split_bf16" > gpurun_out/r1/ops.log 2>&1; echo "ops rc=$?"
timeout 900 python -m pytest tests/test_gpu_speed2d.py -x -q -m gpu > gpurun_out/r1/speed2d_tests.log 2>&1; echo "speed2d tests rc=$?"
for hw in 0 64 256; do
  DEEPHAR_KSPLIT_HW=$hw python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 100 --warmup 20 --dump-steps gpurun_out/r1/steps_speed2d_hw$hw.json > gpurun_out/r1/speed2d_hw$hw.json 2> gpurun_out/r1/speed2d_hw$hw.err; echo "speed2d hw=$hw rc=$?"
done
for hw in 0 256; do
  DEEPHAR_KSPLIT_HW=$hw python bench.py --workload ntu_spnet --no-cpu-baseline --no-predict --steps 20 --warmup 5 --dump-steps gpurun_out/r1/steps_ntu_hw$hw.json > gpurun_out/r1/ntu_hw$hw.json 2> gpurun_out/r1/ntu_hw$hw.err; echo "ntu hw=$hw rc=$?"
  DEEPHAR_KSPLIT_HW=$hw python bench.py --no-cpu-baseline --no-predict --no-bf16x3 --no-clip-leg --steps 30 --warmup 5 --dump-steps gpurun_out/r1/steps_mpii_hw$hw.json > gpurun_out/r1/mpii_hw$hw.json 2> gpurun_out/r1/mpii_hw$hw.err; echo "mpii hw=$hw rc=$?"
done
( time python bench.py --steps 20 --warmup 5 ) > gpurun_out/r1/default_line.json 2> gpurun_out/r1/default_line.err; echo "default rc=$?"
tail -3 gpurun_out/r1/ops.log; tail -3 gpurun_out/r1/speed2d_tests.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r1/*_hw*.json')):
    if 'steps_' in f: continue
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['roofline'].get('whole_forward_frac'))
    except Exception as e: print(f, 'ERR', e)
PY
tail -3 gpurun_out/r1/default_line.err
