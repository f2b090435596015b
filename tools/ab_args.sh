#!/bin/bash
# interleaved comparison of bench.py argument sets on one box: tools/ab_args.sh ROUNDS "args A" "args B" ...
R=$1; shift
cd "$GRAFT_REPO_ROOT"
for i in $(seq $R); do
  for A in "$@"; do
    v=$(timeout 300 python bench.py --no-pmc --no-cpu-baseline $A 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['proof_latency_ms'], d['roofline']['avg_launch_ms'], d['msm_ms'])")
    echo "[$A] $v"
  done
done
