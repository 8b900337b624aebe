# one HIP stream per role (engine/executor.py shared_stream): the predict-then-device slowdown must be gone WITHOUT GPU_MAX_HW_QUEUES,
# nothing else may move; then the tests that exercise streams
one() {
  env $1 python bench.py --workload $2 $3 --no-cpu-baseline --no-predict --no-extra-legs --no-clip-leg --no-bf16x3 --steps $4 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$2 [$3]', d['value'], d['ms_per_step'])"
}
for wl in penn_merge ntu_spnet; do one X=1 $wl "" 30; one X=1 $wl "--pre-predict f32" 30; one X=1 $wl "--pre-predict u8" 30; done
one X=1 speed2d "" 200; one X=1 speed2d "--pre-predict f32" 200
one X=1 mpii "" 30; one X=1 h36m "" 30
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('default line: mpii', d['ms_per_step'], {k:(d[k]['ms_per_step']) for k in ('h36m','ntu_spnet','speed2d','frame_sharded_clips','bf16x3') if k in d}, 'predict fps', d.get('predict_fps_f32'), d.get('predict_fps_u8'))"
python -m pytest tests/test_gpu_nccl.py tests/test_gpu_models.py -q -k "nccl or stream or sharded or predict or pipelin" 2>&1 | tail -3
