cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== bn254 2^16"; bash tools/ab_env.sh 1 "--steps 40 --log-n 16" "APK_MSM_SORT2=0" "APK_MSM_SORT2=1"
echo "== bn254 2^15"; bash tools/ab_env.sh 1 "--steps 40 --log-n 15" "APK_MSM_SORT2=0" "APK_MSM_SORT2=1"
echo "== bls 2^14"; bash tools/ab_env.sh 2 "--steps 40 --curve bls12_381 --log-n 14" "APK_MSM_SORT2=0" "APK_MSM_SORT2=1"
echo "== bn254 2^14"; bash tools/ab_env.sh 1 "--steps 40 --log-n 14" "APK_MSM_SORT2=0" "APK_MSM_SORT2=1"
