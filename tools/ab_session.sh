cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
echo "== two-level sort 0 / auto: bn254 2^17 (value, latency, acc launch, saturated MSM)"
bash tools/ab_env.sh 2 "--steps 40" "APK_MSM_SORT2=0" "APK_MSM_SORT2=-1"
echo "== bn254 2^16"; bash tools/ab_env.sh 2 "--steps 40 --log-n 16" "APK_MSM_SORT2=0" "APK_MSM_SORT2=-1"
echo "== bn254 2^15 (forced)"; bash tools/ab_env.sh 1 "--steps 40 --log-n 15" "APK_MSM_SORT2=0" "APK_MSM_SORT2=1"
