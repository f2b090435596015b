cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== NTT tile / stages under load: bn254 2^17 (value, latency, acc launch, saturated MSM)"
bash tools/ab_env.sh 2 "--steps 40" "APK_NTT_TILE_LOG=9 APK_NTT_STAGES=7" "APK_NTT_TILE_LOG=10 APK_NTT_STAGES=7" "APK_NTT_TILE_LOG=10 APK_NTT_STAGES=9" "APK_NTT_TILE_LOG=11 APK_NTT_STAGES=10"
