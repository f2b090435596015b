for w in 12 13 14 15 16; do
  APK_MSM_WINDOW=$w python bench.py --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('win',$w,'proofs/s',d['value'],'lat_ms',d['proof_latency_ms'],'msm_ms',d['msm_ms'],'acc_ms',d['roofline']['avg_launch_ms'])"
done
