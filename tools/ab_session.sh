cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_time_variants" 2>&1 | tail -3
echo "== bls 2^21 window sweep"
bash tools/ab_env.sh 1 "--curve bls12_381 --log-n 21 --bsb22 1 --steps 5 --warmup 1 --inflight 2" "APK_MSM_WINDOW=16" "APK_MSM_WINDOW=15" "APK_MSM_WINDOW=17"
echo "== bls 2^21 inflight 3/4"
bash tools/ab_args.sh 1 "--curve bls12_381 --log-n 21 --bsb22 1 --steps 5 --warmup 1 --inflight 3" "--curve bls12_381 --log-n 21 --bsb22 1 --steps 4 --warmup 1 --inflight 4"
