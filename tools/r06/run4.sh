mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "grouped or dwconv or pointwise or every_tile or dma_gemm" > gpurun_out/r4/ops.log 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/r4/ops.log
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "grouped or speed_protocol or multi_stream" > gpurun_out/r4/models.log 2>&1; echo "models rc=$?"; tail -3 gpurun_out/r4/models.log
timeout 900 python -m pytest tests/test_gpu_speed2d.py tests/test_gpu_plan_api.py -x -q -m gpu > gpurun_out/r4/speed2d_tests.log 2>&1; echo "speed2d+plan tests rc=$?"; tail -3 gpurun_out/r4/speed2d_tests.log
for g in 0 1 0 1; do
  DEEPHAR_GROUP_LAUNCHES=$g python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 2> gpurun_out/r4/err_$g.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('group=$g', d['value'], d['ms_per_step'], d['config']['streams'])"
done
DEEPHAR_GROUP_LAUNCHES=1 python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 50 --warmup 10 --dump-steps gpurun_out/r4/steps_speed2d.json > /dev/null 2>&1
python bench.py --no-cpu-baseline --no-predict --no-bf16x3 --no-clip-leg --no-extra-legs --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mpii', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
