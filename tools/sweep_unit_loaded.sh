#!/bin/bash
# Accumulate unit length under load (APK_MSM_UNIT_LOADED; 0 = the lone proof's 16..18) across sizes: bench.py lines, two rounds,
# interleaved.  usage: bash tools/sweep_unit_loaded.sh [outdir]
O=${1:-gpurun_out/unit_loaded}
mkdir -p $O; rm -f $O/*.jsonl
one() { tag=$1; u=$2; shift 2; APK_MSM_UNIT_LOADED=$u python bench.py --no-cpu-baseline --no-pmc "$@" 2>/dev/null | tail -1 >> $O/$tag.u$u.jsonl; }
for r in 1 2; do
  for u in 0 24 32 40; do
    one bn254_2p17 $u --steps 20
    one bls12381_2p14 $u --curve bls12_381 --log-n 14 --steps 20
    one bn254_2p15 $u --log-n 15 --steps 20
    one bn254_2p19 $u --log-n 19 --steps 6 --warmup 2
    one bn254_2p17_c17 $u --steps 20 --msm-window 17
  done
done
python - $O <<'PY' | tee $O/summary.txt
import glob, json, os, sys
print("# proofs/s (two interleaved rounds) and lone latency by APK_MSM_UNIT_LOADED; tools/sweep_unit_loaded.sh")
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl"))):
    rows = [json.loads(l) for l in open(f)]
    print("%-28s %-22s lone %s ms  units_by_load %s" % (os.path.basename(f)[:-6], " ".join("%.1f" % r["value"] for r in rows), " ".join("%.3f" % r["proof_latency_ms"] for r in rows),
                                                        rows[0]["paths_under_load"].get("msm_units_by_load")))
PY
