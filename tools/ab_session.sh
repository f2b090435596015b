cd $GRAFT_REPO_ROOT
echo "== unit sweep bn254 2^17"; bash tools/ab_env.sh 2 "--steps 40" "APK_MSM_UNIT=0" "APK_MSM_UNIT=20" "APK_MSM_UNIT=24" "APK_MSM_UNIT=32"
echo "== callers"; bash tools/ab_args.sh 1 "--steps 40 --inflight 24" "--steps 40 --inflight 32" "--steps 30 --inflight 48" "--steps 20 --inflight 64"
echo "== slots"; bash tools/ab_env.sh 1 "--steps 40" "APK_MAX_SLOTS=12" "APK_MAX_SLOTS=16" "APK_MAX_SLOTS=20"
