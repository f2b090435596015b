cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== window 16 vs 17 with the two-level sort: bn254 2^17 (value, latency, acc launch, saturated MSM)"
bash tools/ab_env.sh 2 "--steps 40" "APK_MSM_WINDOW=16" "APK_MSM_WINDOW=17" "APK_MSM_WINDOW=17 APK_MSM_SORT2=1"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "window_sizes" 2>&1 | tail -2
