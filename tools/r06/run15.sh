# what the widened skinny rule (K >= 64, round 5; was K >= 768) did to the bf16x3 mode of the throughput models
for k in 768 64; do for wl in mpii h36m; do for g in f32 bf16x3; do
DEEPHAR_SKINNY_MIN_K=$k python bench.py --workload $wl --gemm $g --no-cpu-baseline --no-predict --no-bf16x3 --no-clip-leg --no-extra-legs --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('min_k=$k $wl $g', d['value'], d['ms_per_step'], 'launches', sum(d['roofline']['kernels_on_main_shape'].values()))"
done; done; done
