# a Model.predict call on host arrays slows the device-resident SPNet-NTU step that follows by 4.7 % -- what about it?
one() {
  env $1 python bench.py --workload $2 $3 --no-cpu-baseline --no-predict --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 $2 [$3]', d['value'], d['ms_per_step'])"
}
one X=1 ntu_spnet ""
one X=1 ntu_spnet "--pre-predict f32"
one GPU_MAX_HW_QUEUES=8 ntu_spnet "--pre-predict f32"
one GPU_MAX_HW_QUEUES=2 ntu_spnet ""
one X=1 ntu_spnet "--pre-predict f32 --no-overlap"
one X=1 ntu_spnet "--no-overlap"
one X=1 h36m ""
one X=1 h36m "--pre-predict f32"
one X=1 penn_merge ""
one X=1 penn_merge "--pre-predict f32"
one X=1 speed2d ""
one X=1 speed2d "--pre-predict f32"
