for w in 14 15 16; do APK_MSM_WINDOW=$w python bench.py --no-cpu-baseline --log-n 20 --steps 2 --warmup 1 --inflight 4 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('bn254 2^20 win',$w,'proofs/s',d['value'],'lat_ms',d['proof_latency_ms'],'msm_ms',d['msm_ms'])"; done
for w in 9 10 11 12 13; do APK_MSM_WINDOW=$w python bench.py --no-cpu-baseline --curve bls12_381 --log-n 14 --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('bls 2^14 win',$w,'proofs/s',d['value'],'lat_ms',d['proof_latency_ms'],'msm_ms',d['msm_ms'])"; done
