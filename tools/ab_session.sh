cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== one-level sort"; APK_MSM_SORT2=0 APK_LIB=$PWD/algoplonk_amd/libapk_knockout.so timeout 300 python tools/knockout.py 17 16 30 2>&1 | tail -13
echo "== two-level sort (skip 1/32/64 have no effect on it: the sort is not skipped)"; APK_MSM_SORT2=1 APK_LIB=$PWD/algoplonk_amd/libapk_knockout.so timeout 300 python tools/knockout.py 17 16 30 2>&1 | grep -E "skip +(0|2|28|125) "
