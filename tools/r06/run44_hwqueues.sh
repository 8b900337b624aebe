# GPU_MAX_HW_QUEUES (ROCm default 4): every workload with 4 / 8 queues, plain and behind a Model.predict call of another model
one() {
  env $1 python bench.py --workload $2 $3 --no-cpu-baseline --no-predict --no-extra-legs --no-clip-leg --no-bf16x3 --steps $4 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 $2 [$3]', d['value'], d['ms_per_step'])"
}
for q in 4 8; do
  for wl in mpii h36m; do one GPU_MAX_HW_QUEUES=$q $wl "" 30; done
  for wl in penn_merge ntu_spnet; do one GPU_MAX_HW_QUEUES=$q $wl "" 30; one GPU_MAX_HW_QUEUES=$q $wl "--pre-predict f32" 30; done
  one GPU_MAX_HW_QUEUES=$q speed2d "" 200; one GPU_MAX_HW_QUEUES=$q speed2d "--pre-predict f32" 200
done
one GPU_MAX_HW_QUEUES=16 ntu_spnet "--pre-predict f32" 30
one GPU_MAX_HW_QUEUES=16 speed2d "" 200
