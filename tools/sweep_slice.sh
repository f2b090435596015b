for sl in 512 1024 2048 4096; do APK_MSM_SLICE=$sl python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('slice',$sl,'proofs/s',d['value'],'lat_ms',d['proof_latency_ms'],'msm_ms',d['msm_ms'])"; done
