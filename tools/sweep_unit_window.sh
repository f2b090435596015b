#!/bin/bash
# Window width x accumulate unit length under load (bench.py lines, interleaved rounds).
# usage: bash tools/sweep_unit_window.sh [outdir] [rounds] [set]      set: mid (2^16..2^19 BN254, default) | small (2^14 BLS12-381, 2^15..2^17 BN254)
O=${1:-gpurun_out/unit_window}
R=${2:-2}
SET=${3:-mid}
mkdir -p $O; rm -f $O/*.jsonl
one() { cvn=$1; lg=$2; c=$3; u=$4; shift 4; APK_MSM_UNIT_LOADED=$u APK_MSM_UNIT_LOADED_BASES=0 python bench.py --no-cpu-baseline --no-pmc --curve $cvn --log-n $lg --msm-window $c "$@" 2>/dev/null | tail -1 >> $O/${cvn}_2p$lg.c$c.u$u.jsonl; }
for r in $(seq $R); do
  if [ $SET = mid ]; then
    one bn254 16 16 0 --steps 20; one bn254 16 16 40 --steps 20; one bn254 16 17 40 --steps 20
    one bn254 17 16 0 --steps 20; one bn254 17 16 40 --steps 20; one bn254 17 17 0 --steps 20; one bn254 17 17 40 --steps 20; one bn254 17 17 64 --steps 20; one bn254 17 18 40 --steps 20
    one bn254 18 17 0 --steps 10; one bn254 18 17 40 --steps 10; one bn254 18 18 40 --steps 10; one bn254 18 18 64 --steps 10
    one bn254 19 17 0 --steps 6 --warmup 2; one bn254 19 17 40 --steps 6 --warmup 2; one bn254 19 18 40 --steps 6 --warmup 2; one bn254 19 18 64 --steps 6 --warmup 2
  else
    for cu in "13 0" "13 32" "14 0" "14 32" "14 48" "15 48"; do set -- $cu; one bls12_381 14 $1 $2 --steps 20; done
    for cu in "15 0" "15 48" "16 48"; do set -- $cu; one bn254 15 $1 $2 --steps 20; done
    for cu in "15 0" "15 48" "16 0" "16 48" "16 64"; do set -- $cu; one bn254 16 $1 $2 --steps 20; done
    for cu in "17 40" "17 48" "17 64"; do set -- $cu; one bn254 17 $1 $2 --steps 20; done
  fi
done
python - $O <<'PY' | tee $O/summary.txt
import glob, json, os, sys
print("# proofs/s (interleaved rounds) and lone latency by window width c and APK_MSM_UNIT_LOADED u (APK_MSM_UNIT_LOADED_BASES=0); tools/sweep_unit_window.sh")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl"))):
    rows = [json.loads(l) for l in open(f)]
    print("%-28s %-26s lone %s ms  sha %s" % (os.path.basename(f)[:-6], " ".join("%.1f" % r["value"] for r in rows), " ".join("%.3f" % r["proof_latency_ms"] for r in rows), rows[0].get("proof_sha256_prefix")))
PY
