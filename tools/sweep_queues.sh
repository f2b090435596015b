for q in 16 24 32; do for f in 16 24 32; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --steps 4 --warmup 1 --inflight $f 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); print('hwq',$q,'inflight',$f,'proofs/s',d['value'],'lat_ms',d['proof_latency_ms'])"
done; done
