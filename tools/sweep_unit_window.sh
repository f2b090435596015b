#!/bin/bash
# Window width x accumulate unit length under load (bench.py lines, interleaved rounds).  usage: bash tools/sweep_unit_window.sh [outdir] [rounds]
O=${1:-gpurun_out/unit_window}
R=${2:-2}
mkdir -p $O; rm -f $O/*.jsonl
one() { lg=$1; c=$2; u=$3; shift 3; APK_MSM_UNIT_LOADED=$u python bench.py --no-cpu-baseline --no-pmc --log-n $lg --msm-window $c "$@" 2>/dev/null | tail -1 >> $O/bn254_2p$lg.c$c.u$u.jsonl; }
for r in $(seq $R); do
  one 16 16 0 --steps 20; one 16 16 40 --steps 20; one 16 17 40 --steps 20
  one 17 16 0 --steps 20; one 17 16 40 --steps 20; one 17 17 0 --steps 20; one 17 17 40 --steps 20; one 17 17 64 --steps 20; one 17 18 40 --steps 20
  one 18 17 0 --steps 10; one 18 17 40 --steps 10; one 18 18 40 --steps 10; one 18 18 64 --steps 10
  one 19 17 0 --steps 6 --warmup 2; one 19 17 40 --steps 6 --warmup 2; one 19 18 40 --steps 6 --warmup 2; one 19 18 64 --steps 6 --warmup 2
done
python - $O <<'PY' | tee $O/summary.txt
import glob, json, os, sys
print("# proofs/s (interleaved rounds) and lone latency by window width c and APK_MSM_UNIT_LOADED u; tools/sweep_unit_window.sh")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl"))):
    rows = [json.loads(l) for l in open(f)]
    print("%-28s %-26s lone %s ms  sha %s" % (os.path.basename(f)[:-6], " ".join("%.1f" % r["value"] for r in rows), " ".join("%.3f" % r["proof_latency_ms"] for r in rows), rows[0].get("proof_sha256_prefix")))
PY
