#!/bin/bash
# Collects what profiles/ keeps for a round (run through gpurun from the repo root): bench lines, kernel traces, PMC passes.
# usage: tools/profile_round.sh TAG     -> gpurun_out/TAG_*
TAG=${1:-rXX}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 400 python bench.py > $O/${TAG}_bench_bn254_2p17.log 2>&1; tail -1 $O/${TAG}_bench_bn254_2p17.log > $O/${TAG}_bench_bn254_2p17.json
timeout 300 python bench.py --curve bls12_381 --log-n 14 > $O/${TAG}_bench_bls12381_2p14.log 2>&1; tail -1 $O/${TAG}_bench_bls12381_2p14.log > $O/${TAG}_bench_bls12381_2p14.json
timeout 300 python bench.py --curve bls12_381 --log-n 21 --bsb22 1 --inflight 4 --steps 4 --warmup 1 --no-pmc --no-cpu-baseline > $O/${TAG}_bench_bls12381_2p21_bsb22.log 2>&1; tail -1 $O/${TAG}_bench_bls12381_2p21_bsb22.log > $O/${TAG}_bench_bls12381_2p21_bsb22.json
timeout 200 python bench.py --mode msm-sharded --steps 50 > $O/${TAG}_bench_msm_sharded.log 2>&1; tail -1 $O/${TAG}_bench_msm_sharded.log > $O/${TAG}_bench_msm_sharded.json
timeout 200 python bench.py --mode prove-split --curve bls12_381 --log-n 21 --steps 5 --warmup 1 > $O/${TAG}_bench_prove_split_2p21.log 2>&1; tail -1 $O/${TAG}_bench_prove_split_2p21.log > $O/${TAG}_bench_prove_split_2p21.json
# kernel traces: one proof at a time (sequential) and the bench's 32 callers (saturated)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt1 -o r -- python bench.py --inflight 1 --steps 8 --warmup 2 --no-cpu-baseline --no-pmc > $O/${TAG}_kt1.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_kt1/r_results.db > $O/${TAG}_kernel_trace_bn254_2p17.txt
grep '"metric"' $O/${TAG}_kt1.log | tail -1 > $O/${TAG}_kernel_trace_bn254_2p17_benchline.json
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt24 -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc > $O/${TAG}_kt24.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_kt24/r_results.db > $O/${TAG}_kernel_trace_bn254_2p17_saturated.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_ktbls -o r -- python bench.py --curve bls12_381 --log-n 14 --inflight 1 --steps 8 --warmup 2 --no-cpu-baseline --no-pmc > $O/${TAG}_ktbls.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_ktbls/r_results.db > $O/${TAG}_kernel_trace_bls12381_2p14.txt
# PMC passes (one counter group per pass, kernel trace only: no other trace domains)
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/${TAG}_pmc_$N -o p -- python tools/prof_msm.py 17 4 2 > $O/${TAG}_pmc_$N.log 2>&1
  python tools/pmc_summary.py $O/${TAG}_pmc_$N/p_results.db > $O/${TAG}_pmc_bn254_2p17_$N.txt
done
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/${TAG}_pmcbls_$N -o p -- python tools/prof_msm.py 14 4 2 bls12_381 > $O/${TAG}_pmcbls_$N.log 2>&1
  python tools/pmc_summary.py $O/${TAG}_pmcbls_$N/p_results.db > $O/${TAG}_pmc_bls12381_2p14_$N.txt
done
rm -rf $O/${TAG}_kt1 $O/${TAG}_kt24 $O/${TAG}_ktbls $O/${TAG}_pmc_*/ $O/${TAG}_pmcbls_*/
ls -la $O | grep ${TAG} | tail -40
