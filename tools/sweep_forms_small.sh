#!/bin/bash
# The load-dependent kernel forms, one at a time forced the other way, on BLS12-381 2^14 under load (bench.py lines, two interleaved
# rounds).  usage: bash tools/sweep_forms_small.sh [outdir] [curve] [log_n]
O=${1:-gpurun_out/forms_small}; CV=${2:-bls12_381}; LG=${3:-14}
mkdir -p $O; rm -f $O/*.jsonl
b() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-pmc --curve $CV --log-n $LG --steps 20 2>/dev/null | tail -1 >> $O/$tag.jsonl; }
for r in 1 2; do
  b default APK_NOTHING=1
  b lean_tail_off APK_MSM_LEAN_TAIL=0
  b rowcol_serial_off APK_MSM_ROWCOL_SERIAL=0
  b sort2_off APK_MSM_SORT2=0
  b radix4_on APK_NTT_RADIX4=1
  b combine_quad_on APK_MSM_COMBINE_QUAD=1
  b quad_tail_7 APK_MSM_QUAD_TAIL=7
  b quad_tail_0 APK_MSM_QUAD_TAIL=0
  b blocking_sync_off APK_SYNC_BLOCKING=0
  b unit_24 APK_MSM_UNIT=24
  b unit_small_waves4 APK_MSM_SMALL_WAVES=4
done
python - $O <<'PY' | tee $O/summary.txt
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl"))):
    rows = [json.loads(l) for l in open(f) if l.strip().startswith("{")]
    print("%-22s %-20s lone %s" % (os.path.basename(f)[:-6], " ".join("%.1f" % r["value"] for r in rows), " ".join("%.3f" % r["proof_latency_ms"] for r in rows)))
PY
