#!/bin/bash
# interleaved comparison of several builds of libapk on one box: tools/ab_libs.sh ROUNDS "bench args" lib1.so lib2.so ...
R=$1; ARGS=$2; shift 2
cd "$GRAFT_REPO_ROOT"
for i in $(seq $R); do
  for L in "$@"; do
    v=$(APK_LIB=$PWD/$L timeout 300 python bench.py --no-pmc --no-cpu-baseline $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['proof_latency_ms'], d['roofline']['avg_launch_ms'], d.get('msm_mscalar_per_s_saturated'), d.get('ntt_ms_per_proof'))")
    echo "$L $v"
  done
done
