#!/bin/bash
# interleaved comparison of environment settings on one box: tools/ab_env.sh ROUNDS "bench args" "ENV1=.. ENV2=.." "ENV.." ...
R=$1; ARGS=$2; shift 2
cd "$GRAFT_REPO_ROOT"
for i in $(seq $R); do
  for E in "$@"; do
    v=$(env $E timeout 300 python bench.py --no-pmc --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['proof_latency_ms'], d['roofline']['avg_launch_ms'], d['msm_mscalar_per_s_saturated'])")
    echo "[$E] $v"
  done
done
