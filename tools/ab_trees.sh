#!/bin/bash
# Same-box, interleaved comparison of two CHECKOUTS of the repo (each with its own built libapk.so and bench.py) - for comparing
# rounds whose C-ABI differs (tools/ab_libs.sh swaps only the library under one bench.py).
# usage: tools/ab_trees.sh ROUNDS "bench args" treeA treeB ...      (trees relative to the repo root; "." = this tree)
# AB_ARGS_HERE: extra arguments for THIS tree only (flags an older bench.py does not know); AB_TIMEOUT: seconds per run (400).
# The other tree: `git worktree add ab/r04 <commit> && make -C ab/r04/algoplonk_amd/csrc -j3 && make -C ab/r04/oracle` (ab/ is
# git-ignored and travels to the GPU box with the snapshot; `git worktree remove ab/r04` afterwards).
R=$1; ARGS=$2; shift 2
cd "$GRAFT_REPO_ROOT"
for i in $(seq $R); do
  for T in "$@"; do
    EX=""; [ "$T" = "." ] && EX="$AB_ARGS_HERE"
    v=$(cd $T && timeout ${AB_TIMEOUT:-400} python bench.py --no-pmc --no-cpu-baseline $ARGS $EX 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['proof_latency_ms'], d['roofline']['avg_launch_ms'], d.get('msm_mscalar_per_s_saturated'), d.get('ntt_ms_per_proof'), d.get('msm_ms'))")
    echo "$T $v"
  done
done
