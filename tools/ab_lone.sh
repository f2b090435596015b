#!/bin/bash
# Same-box A/B of a lone proof's latency across the host-side knobs of the prover's round 4 (the [lin] combination between the
# evaluations and gamma'): tools/lone_latency.py, one process per setting, the settings interleaved, two rounds; then a table.
# usage: bash tools/ab_lone.sh [outdir]        (the table goes to $outdir/summary.txt)
O=${1:-gpurun_out/ab_lone}
mkdir -p $O
rm -f $O/*.jsonl
run() { n=$1; lg=$2; cvn=$3; shift 3; env "$@" python tools/lone_latency.py $lg $cvn 40 16 >> $O/$n.$cvn.jsonl 2>>$O/err.log; }
for rep in 1 2; do
 for c in "17 bn254" "14 bls12_381"; do
  set -- $c
  run 0_round4_form $1 $2 APK_HOST_GLV=0 APK_HOST_FIXED=0 APK_LIN_EARLY_H=0 APK_HOST_LINCOMB_THREADS=4
  run 1_default $1 $2 APK_NOTHING=1
  run 2_no_early_h $1 $2 APK_LIN_EARLY_H=0
  run 3_no_fixed_base $1 $2 APK_HOST_FIXED=0
  run 4_no_glv $1 $2 APK_HOST_GLV=0
  run 5_one_thread $1 $2 APK_HOST_LINCOMB_THREADS=1
  run 6_three_threads $1 $2 APK_HOST_LINCOMB_THREADS=3
  run 7_six_threads $1 $2 APK_HOST_LINCOMB_THREADS=6
 done
done
python - $O <<'PY' > $O/summary.txt
import glob, json, os, sys
print("# lone proof on a 16-slot context, 40 proofs per process, two interleaved rounds per setting (tools/ab_lone.sh)")
print("# median_ms / min_ms: wall clock of apk_prove_device; lincomb_ms: the library's own timer around the [lin] combination (instrumented proofs)")
print("%-12s %-18s %-17s %-17s %-17s %s" % ("curve", "setting", "median_ms", "min_ms", "lincomb_ms", "sha256"))
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl")), key=lambda p: (p.split(".")[-2], p)):
    rows = [json.loads(l) for l in open(f)]
    name, cv = os.path.basename(f).split(".")[:2]
    print("%-12s %-18s %-17s %-17s %-17s %s" % (cv, name, " ".join("%.3f" % r["median_ms"] for r in rows), " ".join("%.3f" % r["min_ms"] for r in rows),
                                          " ".join("%.3f" % r["instrumented"]["host_lincomb_ms"] for r in rows), rows[0]["sha256"]))
PY
cat $O/summary.txt
