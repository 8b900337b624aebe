# pooled epilogue at 16 / 8 columns (R7 small) + R13: tests, then same-box A/B
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "pooled_second_output" 2>&1 | tail -15
python -m pytest tests/test_gpu_models.py -x -q -k "sibling_pools or pooled_epilogue" 2>&1 | tail -15
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --no-extra-legs --steps $3 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 $2', d['value'], d['ms_per_step'])"
}
for rep in 1 2 3; do
one DEEPHAR_FUSE_POOL_SMALL=0 speed2d 200
one DEEPHAR_FUSE_POOL_SMALL=1 speed2d 200
done
for wl in mpii h36m penn_merge ntu_spnet; do
one DEEPHAR_FUSE_POOL_SMALL=0 $wl 30
one DEEPHAR_FUSE_POOL_SMALL=1 $wl 30
done
