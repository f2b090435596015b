mkdir -p gpurun_out/r6d
run() { # curve logn gang inflight
  env APK_GANG=$3 $5 python bench.py --curve $1 --log-n $2 --inflight $4 --steps ${STEPS:-20} --warmup 3 --no-pmc --no-cpu-baseline --no-host-inputs --no-oracle-check > gpurun_out/r6d/b_$1_$2_g$3_i$4.json 2> gpurun_out/r6d/err.txt || tail -3 gpurun_out/r6d/err.txt
  python - <<PY
import json
d=json.load(open("gpurun_out/r6d/b_$1_$2_g$3_i$4.json"))
p=d["paths_under_load"]
print("$1 2^$2 gang=$3 inflight=$4 $5: %.1f proofs/s  lone %.0f us  gang_proofs %d/%d msm_batches %d ok=%s cpu=%s" % (d["value"], d["proof_latency_ms"]*1000+d["ntt_ms_per_proof"]*0, p["gang_proofs"], p["proofs"], p["msm_batches"], d["proofs_under_load_match_lone_proofs"], {"ntt_ms_per_proof": d["ntt_ms_per_proof"]}))
PY
}
for rep in 1 2; do
run bn254 17 1 32 APK_NTT_TWU=0
run bn254 17 1 32 APK_NTT_TWU=1
run bls12_381 14 4 64 APK_NTT_TWU=0
run bls12_381 14 4 64 APK_NTT_TWU=1
done
STEPS=6; run bls12_381 21 1 4 APK_NTT_TWU=0
run bls12_381 21 1 4 APK_NTT_TWU=1
