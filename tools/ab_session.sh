cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
echo "== scatter pass with the stores two windows behind the atomics: bn254 2^17 (value, latency, acc launch, saturated MSM, ntt)"
bash tools/ab_libs.sh 3 "--steps 40" algoplonk_amd/libapk_nopipe.so algoplonk_amd/libapk.so
