# PMC passes over the in-model launches of SPNet-NTU's first-layer kernel (step 0) and its 3x3 halo-resident convolutions
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_halo; mkdir -p $O
for step in 0 3 13; do
 for grp in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-trace -d $O/s${step}_$tag -o out --output-format csv -- python $R/bench.py --workload ntu_spnet --tune-cache /tmp/tune_ntu.json --no-cpu-baseline --no-predict --replay-step $step --replay-reps 4 > $O/log_${step}_$tag.txt 2>&1
  f=$(find $O/s${step}_$tag -name "*counter_collection.csv" | head -1)
  python - "$f" $step "$tag" <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
# last 4 dispatches of the replayed kernel: take the kernel name of the last row
last=rows[-1]['Kernel_Name']
sel=[r for r in rows if r['Kernel_Name']==last]
ids=sorted({int(r['Dispatch_Id']) for r in sel})[-4:]
agg=collections.defaultdict(float)
for r in sel:
    if int(r['Dispatch_Id']) in ids: agg[r['Counter_Name']]+=float(r['Counter_Value'])
print('step', sys.argv[2], last[:60], {k: round(v/len(ids),1) for k,v in agg.items()})
PY
  rm -rf $O/s${step}_$tag
 done
done
