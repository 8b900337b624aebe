python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "first_layer or uint8 or u8" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_stem; mkdir -p $O
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/s0 -o out --output-format csv -- python $R/bench.py --workload ntu_spnet --no-cpu-baseline --no-predict --replay-step 0 --replay-reps 4 > $O/log.txt 2>&1
f=$(find $O/s0 -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
last=rows[-1]['Kernel_Name']; sel=[r for r in rows if r['Kernel_Name']==last]
ids=sorted({int(r['Dispatch_Id']) for r in sel})[-4:]
agg=collections.defaultdict(float)
for r in sel:
    if int(r['Dispatch_Id']) in ids: agg[r['Counter_Name']]+=float(r['Counter_Value'])
a={k:v/len(ids) for k,v in agg.items()}
print(last[:60], {k:round(v,1) for k,v in a.items()}, 'mfma busy %.3f'%(a['SQ_VALU_MFMA_BUSY_CYCLES']/1024/(a['GRBM_GUI_ACTIVE']/8)), 'conflict share %.3f'%(a['SQ_LDS_BANK_CONFLICT']/a['SQ_LDS_IDX_ACTIVE']))
PY
rm -rf $O/s0
cd $R
python bench.py --workload ntu_spnet --no-cpu-baseline --no-predict --steps 20 --warmup 5 --dump-steps gpurun_out/steps_ntu_stem.json 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ntu', d['value'], d['ms_per_step'], d['roofline'].get('whole_forward_frac'))"
python -c "
import json; d=json.load(open('gpurun_out/steps_ntu_stem.json')); print('conv1', round(d[0]['ms']*1e3,1), d[0]['kernel'])"
