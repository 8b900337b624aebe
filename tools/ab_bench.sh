#!/bin/bash
# A/B of planner / library switches on the GPU box: tools/ab_bench.sh <tag> <workload> [ENV=VAL ...] -> one summary line
TAG=$1; W=$2; shift 2
OUT=${GRAFT_REPO_ROOT:-.}/gpurun_out
env "$@" python bench.py --workload $W --steps 30 --warmup 5 --no-cpu-baseline --no-predict --no-clip-leg --no-bf16x3 \
    --dump-steps $OUT/${TAG}_steps_$W.json > $OUT/${TAG}_line_$W.json 2> $OUT/${TAG}_err_$W.log
python - <<PY
import json
try:
    d = json.loads(open("$OUT/${TAG}_line_$W.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print("$TAG $W $*", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["whole_forward_frac"])
except Exception as e:
    print("$TAG $W FAILED", e); print(open("$OUT/${TAG}_err_$W.log").read()[-1500:])
PY
