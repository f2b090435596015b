# the round's bench lines (driver arguments = bench.py's defaults) into gpurun_out/lines/: bash tools/round_lines.sh
mkdir -p gpurun_out/lines
python bench.py > gpurun_out/lines/bench_bn254_2p17.json 2> gpurun_out/lines/err_bn254_2p17.txt || tail -5 gpurun_out/lines/err_bn254_2p17.txt
python bench.py --curve bls12_381 --log-n 14 > gpurun_out/lines/bench_bls12381_2p14.json 2> gpurun_out/lines/err_bls14.txt || tail -5 gpurun_out/lines/err_bls14.txt
python bench.py --curve bls12_381 --log-n 14 --inflight 64 > gpurun_out/lines/bench_bls12381_2p14_64callers.json 2> gpurun_out/lines/err_bls14_64.txt || tail -5 gpurun_out/lines/err_bls14_64.txt
python - <<PY
import json
for f in ("bench_bn254_2p17", "bench_bls12381_2p14", "bench_bls12381_2p14_64callers"):
    try:
        d = json.load(open("gpurun_out/lines/%s.json" % f))
    except Exception as e:
        print(f, "unreadable", e); continue
    rf = d["roofline"]
    print(f, d["value"], "host", d["value_host_inputs"], d["value_host_inputs_pageable"], "lat", d["proof_latency_ms"], d["proof_latency_host_inputs_ms"],
          "oracle", all(d["matches_oracle"]) if d["matches_oracle"] else d["matches_oracle"], "frac", rf["frac"], "valu", (rf.get("valu") or {}).get("frac"),
          "valu_under_load", rf.get("valu_under_load"), "proof", rf["proof"]["frac"], "ntt", rf["ntt"]["frac"], "cpu", d["cpu_baseline"]["value"], d["host_cpu_timed_region"])
PY
