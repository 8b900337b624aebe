# experiment: skinny-conv kernel also for wide layers (Cout > 256) on maps of at most DEEPHAR_SKINNY_WIDE_PX positions
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --no-extra-legs --no-clip-leg --no-bf16x3 --steps $3 --warmup 10 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 $2', d['value'], d['ms_per_step'], d['roofline'].get('launches_per_forward'))"
}
for px in 0 16 64 256 0; do one DEEPHAR_SKINNY_WIDE_PX=$px speed2d 200; done
for wl in mpii h36m penn_merge ntu_spnet; do for px in 0 16 64 256; do one DEEPHAR_SKINNY_WIDE_PX=$px $wl 30; done; done
