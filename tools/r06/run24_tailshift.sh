# where the suffix of the 'tail' policy should start now that launches are merged at bind time: sweep of DEEPHAR_TAIL_SHIFT
one() {
  env $1 python bench.py --workload speed2d --no-cpu-baseline --no-predict --no-extra-legs --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1', d['value'], d['ms_per_step'], d['roofline'].get('launches_per_forward'))"
}
for sh in 0 -20 -10 10 20 30 45 60 80 0; do one DEEPHAR_TAIL_SHIFT=$sh; done
