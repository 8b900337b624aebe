# which part of the default line slows the compact ntu_spnet leg by 4 %?
for flags in "--no-predict" "--no-cpu-baseline" "--no-bf16x3" "--no-clip-leg" ""; do
python bench.py $flags --steps 10 --warmup 3 2>/dev/null | python -c "
import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('flags [$flags]', 'mpii', d['ms_per_step'], {k:(d[k]['ms_per_step']) for k in ('h36m','ntu_spnet','speed2d') if k in d})"
done
