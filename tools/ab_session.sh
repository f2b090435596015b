cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
echo "== previous / new second-level tile (value, latency, acc, sat MSM, ntt)"
bash tools/ab_libs.sh 2 "--steps 30" algoplonk_amd/libapk_prev.so algoplonk_amd/libapk.so
