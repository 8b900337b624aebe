# XCD-aware work-group order of the decoder kernels: tests, per-launch PMC of the fused decoder, same-box A/B against a library
# built with -DDH_DECODER_XCD_ORDER=0 (deephar_amd/csrc/build/variant_decoder_noxcd.so)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -q -k "softargmax or sam or decoder or context or depth or kron" 2>&1 | tail -3
python -m pytest tests/test_gpu_models.py -q -k "golden or oracle or decoder" 2>&1 | tail -3
python tools/pmc_all_kernels.py mpii 2>&1 | grep -E "^==|sam_ctx" | head -4
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --no-extra-legs --no-clip-leg --no-bf16x3 --steps $3 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 $2', d['value'], d['ms_per_step'])"
}
V=DEEPHAR_HIP_LIB=$PWD/deephar_amd/csrc/build/variant_decoder_noxcd.so
for rep in 1 2; do one $V mpii 40; one X=1 mpii 40; done
one $V h36m 30; one X=1 h36m 30
one $V penn_merge 30; one X=1 penn_merge 30
one $V ntu_spnet 20; one X=1 ntu_spnet 20
one $V speed2d 200; one X=1 speed2d 200
