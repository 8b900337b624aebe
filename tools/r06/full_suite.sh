mkdir -p gpurun_out
( time python -m pytest tests/ -q -m gpu ) > gpurun_out/full_suite.log 2>&1; echo "suite rc=$?"; tail -6 gpurun_out/full_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for wl in penn_merge ntu_spnet; do
  python bench.py --workload $wl --force-collective --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/forced_$wl.json 2> gpurun_out/forced_$wl.err; echo "forced $wl rc=$?"
  python - "$wl" <<'PY'
import json,sys
d=json.loads([l for l in open('gpurun_out/forced_%s.json'%sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print({k:d.get(k) for k in ('value','ms_per_step','collective_us','serial_form_ms_per_step','hidden_by_pipelining_us','per_rank_ms_min_max')})
PY
done
