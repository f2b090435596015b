#!/bin/bash
# A/B of two builds of libapk on the same box, interleaved: tools/ab_bench.sh A.so B.so [rounds] [bench args...]
A=$1; B=$2; R=${3:-3}; shift 3
cd "$GRAFT_REPO_ROOT"
for i in $(seq $R); do
  for L in $A $B; do
    v=$(APK_LIB=$PWD/$L timeout 300 python bench.py --no-pmc --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['proof_latency_ms'], d['roofline']['avg_launch_ms'])")
    echo "$L $v"
  done
done
