#!/bin/bash
# The defaults that came out of tools/sweep_unit_window.sh against the forms they replaced, two interleaved rounds of bench.py lines
# per setting (2^16 / 2^17 / 2^18 BN254, 2^14 / 2^17 BLS12-381).  usage: bash tools/verify_unit_defaults.sh   -> gpurun_out/verify_unit/summary.txt
O=gpurun_out/verify_unit; mkdir -p $O; rm -f $O/*.jsonl
b() { tag=$1; shift; env "$@" 2>/dev/null | tail -1 >> $O/$tag.jsonl; }
for r in 1 2; do
  b bn16_default python bench.py --no-cpu-baseline --no-pmc --log-n 16 --steps 20
  b bn16_c15_u0 APK_MSM_UNIT_LOADED=0 python bench.py --no-cpu-baseline --no-pmc --log-n 16 --steps 20 --msm-window 15
  b bn17_default python bench.py --no-cpu-baseline --no-pmc --steps 20
  b bn17_c16_u0 APK_MSM_UNIT_LOADED=0 python bench.py --no-cpu-baseline --no-pmc --steps 20 --msm-window 16
  b bls17_default python bench.py --no-cpu-baseline --no-pmc --curve bls12_381 --log-n 17 --steps 12
  b bls17_u0 APK_MSM_UNIT_LOADED=0 python bench.py --no-cpu-baseline --no-pmc --curve bls12_381 --log-n 17 --steps 12
  b bls17_c17 python bench.py --no-cpu-baseline --no-pmc --curve bls12_381 --log-n 17 --steps 12 --msm-window 17
  b bls14_default python bench.py --no-cpu-baseline --no-pmc --curve bls12_381 --log-n 14 --steps 20
  b bn18_default python bench.py --no-cpu-baseline --no-pmc --log-n 18 --steps 10
  b bn18_u0 APK_MSM_UNIT_LOADED=0 python bench.py --no-cpu-baseline --no-pmc --log-n 18 --steps 10
done
python - <<'PY' | tee $O/summary.txt
import glob, json, os
for f in sorted(glob.glob("gpurun_out/verify_unit/*.jsonl")):
    rows = [json.loads(l) for l in open(f)]
    print("%-16s %-22s lone %s ms  c=%s units_by_load %s sha %s" % (os.path.basename(f)[:-6], " ".join("%.1f" % r["value"] for r in rows), " ".join("%.3f" % r["proof_latency_ms"] for r in rows), rows[0]["msm_window"], rows[0]["paths_under_load"].get("msm_units_by_load"), rows[0].get("proof_sha256_prefix")))
PY
