cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
echo "== XCD-aware slice order off/on: bn254 2^17"
bash tools/ab_env.sh 3 "--steps 40" "APK_MSM_XCD_ORDER=0" "APK_MSM_XCD_ORDER=1"
echo "== bls 2^14"; bash tools/ab_env.sh 2 "--steps 40 --curve bls12_381 --log-n 14" "APK_MSM_XCD_ORDER=0" "APK_MSM_XCD_ORDER=1"
echo "== bls 2^21"; bash tools/ab_env.sh 1 "--curve bls12_381 --log-n 21 --bsb22 1 --steps 4 --warmup 1 --inflight 4" "APK_MSM_XCD_ORDER=0" "APK_MSM_XCD_ORDER=1"
