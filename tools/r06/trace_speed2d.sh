# kernel timeline of the replayed speed2d graph (one- and two-stream), for the critical-path analysis of tools/r06/analyze_trace.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/trace
for st in 1 2; do
  DEEPHAR_KSPLIT_HW=${KSHW:-0} rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace/s$st -- python $R/bench.py --workload speed2d --no-cpu-baseline --no-predict --steps 6 --warmup 3 --streams $st --stream-policy tail --tune-cache /tmp/tune_speed2d.json > $R/gpurun_out/trace/line_s$st.json 2> $R/gpurun_out/trace/err_s$st.txt
  echo "streams=$st rc=$?"
done
cd $R/gpurun_out/trace && for st in 1 2; do f=$(find s$st -name "*kernel_trace.csv" | head -1); echo $f; python - "$f" s$st <<'PY'
import csv, sys, gzip, json
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# keep the last ~620*3 kernels, compress to a small json: name, start, end, queue, stream
keep=rows[-6500:]
t0=int(keep[0]['Start_Timestamp'])
out=[dict(n=r['Kernel_Name'][:70], s=int(r['Start_Timestamp'])-t0, e=int(r['End_Timestamp'])-t0, q=r.get('Queue_Id'), st=r.get('Stream_Id'), wg=r.get('Workgroup_Size_X'), gx=r.get('Grid_Size_X')) for r in keep]
json.dump(out, open(sys.argv[2]+'_tail.json','w'))
print(len(rows), 'kernels; columns', list(rows[0].keys()))
PY
done
rm -rf s1 s2
