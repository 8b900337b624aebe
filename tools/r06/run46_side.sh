# side streams of multi-stream plans = the engine's role streams: stream tests, then the speed protocol
python -m pytest tests/test_gpu_speed2d.py tests/test_gpu_models.py -q -k "stream or speed2d or tail or grouped or paired" 2>&1 | tail -3
python bench.py --workload speed2d --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step']); print(d['speed2d']['fps_per_block']); print(d['speed2d']['fps_per_block_protocol_exact'])"
python bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 200 --warmup 20 --pre-predict f32 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('behind predict', d['value'], d['ms_per_step'])"
