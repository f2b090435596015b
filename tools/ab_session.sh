cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in knockout sc1 sc2 sc3; do
  echo "== $L"
  APK_LIB=$PWD/algoplonk_amd/libapk_$L.so timeout 300 python tools/knockout.py 17 16 30 2>&1 | grep -E "skip +(0|64|97) "
done
