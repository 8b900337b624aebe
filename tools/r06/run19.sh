# R13 (sibling pools out of a joint buffer) same-box A/B on the speed protocol's last model + the pre-split K-loop model
mkdir -p gpurun_out
python -m pytest tests/test_gpu_models.py -x -q -k "sibling_pools or resample_on_load" 2>&1 | tail -3
one() {
  env $1 python bench.py --workload $2 --no-cpu-baseline --no-predict --steps $3 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', d['value'], d['ms_per_step'], d['roofline'].get('launches'))"
}
for rep in 1 2 3; do
one DEEPHAR_MERGE_POOLS=0 speed2d 200
one DEEPHAR_MERGE_POOLS=1 speed2d 200
done
one DEEPHAR_MERGE_POOLS=0 penn_merge 30
one DEEPHAR_MERGE_POOLS=1 penn_merge 30
one DEEPHAR_MERGE_POOLS=0 ntu_spnet 20
one DEEPHAR_MERGE_POOLS=1 ntu_spnet 20
bash tools/r06/run18_presplit_model.sh
