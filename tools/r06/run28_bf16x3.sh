# the split-bf16 wide tiling without register spills in its epilogue (conv_epilogue CHROWS = 6): tests, then same-box A/B against
# a library built with -DDH_WIDE_CHROWS=0 (deephar_amd/csrc/build/variant_wide_chrows0.so: the previous code)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16x3.py -q 2>&1 | tail -3
python -m pytest tests/test_gpu_ops.py -q -k "split or bf16 or pooled_second" 2>&1 | tail -3
one() {
  env $1 python bench.py --workload $2 --gemm bf16x3 --no-cpu-baseline --no-predict --no-extra-legs --no-clip-leg --steps 30 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('$1 $2', d['value'], d['ms_per_step'], r['kernel'], r.get('main_shape_avg_launch_us'))"
}
V=DEEPHAR_HIP_LIB=$PWD/deephar_amd/csrc/build/variant_wide_chrows0.so
for rep in 1 2 3; do
one $V mpii
one X=1 mpii
done
one $V h36m
one X=1 h36m
