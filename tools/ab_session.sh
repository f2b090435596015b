cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
echo "== merge spill off/on: bn254 2^17"
bash tools/ab_env.sh 3 "--steps 40" "APK_MSM_SPILL=0" "APK_MSM_SPILL=1"
echo "== bls 2^14 window"
bash tools/ab_env.sh 1 "--steps 40 --curve bls12_381 --log-n 14" "APK_MSM_WINDOW=13" "APK_MSM_WINDOW=14"
echo "== bn254 2^16 window"
bash tools/ab_env.sh 1 "--steps 40 --log-n 16" "APK_MSM_WINDOW=14" "APK_MSM_WINDOW=15" "APK_MSM_WINDOW=16"
echo "== bn254 2^15 window"
bash tools/ab_env.sh 1 "--steps 40 --log-n 15" "APK_MSM_WINDOW=13" "APK_MSM_WINDOW=14" "APK_MSM_WINDOW=15"
