# Gangs: proofs/s by size, gang size (APK_GANG) and callers - short bench.py runs, parity of the timed region checked by each.
# usage (through gpurun, from the repo root): bash tools/sweep_gangs.sh
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
# the sweep behind profiles/r06_gang_sweep.txt sections 3-5 (edit the list for other questions: section 6 and r06_ntt_twiddles.txt were
# made with `run <curve> <log_n> <gang> <callers> "ENV=VALUE ..."` lines of their own)
for cfg in "bn254 13" "bn254 15" "bls12_381 14"; do
  set -- $cfg
  run $1 $2 1 32
  run $1 $2 2 32
  run $1 $2 4 64
done
run bn254 16 1 32
run bn254 16 2 32
run bn254 17 1 32
run bn254 17 2 32
run bls12_381 14 4 64 APK_GANG_DEFER=0
