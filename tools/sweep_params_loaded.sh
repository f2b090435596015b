#!/bin/bash
# Sort / NTT / tail parameters under load at BN254 2^17 (17-bit windows), one at a time off their defaults (bench.py lines, two
# interleaved rounds).  usage: bash tools/sweep_params_loaded.sh [outdir]
O=${1:-gpurun_out/params_loaded}
mkdir -p $O; rm -f $O/*.jsonl
b() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-pmc --steps 20 2>/dev/null | tail -1 >> $O/$tag.jsonl; }
for r in 1 2; do
  b default APK_NOTHING=1
  b part_target_8192 APK_MSM_PART_TARGET=8192
  b part_target_4096 APK_MSM_PART_TARGET=4096
  b part_pblog_4 APK_MSM_PART_PBLOG=4
  b part_pblog_6 APK_MSM_PART_PBLOG=6
  b slice_1024 APK_MSM_SLICE=1024
  b slice_3072 APK_MSM_SLICE=3072
  b rowcol_lanes_8 APK_MSM_ROWCOL_LANES=8
  b sorted_merge_off APK_MSM_SORTED_MERGE=0
  b combine_dyn_off APK_MSM_COMBINE_DYN=0
  b ntt_tile_10 APK_NTT_TILE_LOG=10
  b ntt_tile_11 APK_NTT_TILE_LOG=11
  b ntt_threads_128 APK_NTT_THREADS=128
  b radix4_off APK_NTT_RADIX4=0
  b unit_loaded_64 APK_MSM_UNIT_LOADED=64
done
python - $O <<'PY' | tee $O/summary.txt
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.jsonl"))):
    rows = [json.loads(l) for l in open(f) if l.strip().startswith("{")]
    print("%-22s %-20s lone %s" % (os.path.basename(f)[:-6], " ".join("%.1f" % r["value"] for r in rows), " ".join("%.3f" % r["proof_latency_ms"] for r in rows)))
PY
