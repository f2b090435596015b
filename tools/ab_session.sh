cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== default rule / two-level sort for every batch size (value, latency, acc, sat MSM, then msm_ms)"
for i in 1 2; do for E in "APK_MSM_SORT2=-1" "APK_MSM_SORT2=1"; do
  v=$(env $E timeout 300 python bench.py --no-pmc --no-cpu-baseline --steps 12 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['proof_latency_ms'], d['msm_ms'], d['msm_mscalar_per_s_saturated'])")
  echo "[$E] $v"; done; done
