mkdir -p gpurun_out
python -m pytest tests/test_gpu_models.py -x -q -s -k "paired_skinny or grouped_launches" 2>&1 | grep -v amdgpu.ids | tail -15
python bench.py --workload speed2d --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('launches_per_forward'), d['roofline'].get('plan_steps')); print(d['speed2d']['launches_per_call']); print(d['speed2d']['fps_per_block'])"
