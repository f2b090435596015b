#!/bin/bash
# Collects what profiles/ keeps for a round (run through gpurun from the repo root): bench lines, kernel traces, PMC passes.
# usage: tools/profile_round.sh TAG     -> gpurun_out/TAG_*
TAG=${1:-rXX}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python bench.py > $O/${TAG}_bench_bn254_2p17.log 2>&1; tail -1 $O/${TAG}_bench_bn254_2p17.log > $O/${TAG}_bench_bn254_2p17.json
python bench.py --curve bls12_381 --log-n 14 > $O/${TAG}_bench_bls12381_2p14.log 2>&1; tail -1 $O/${TAG}_bench_bls12381_2p14.log > $O/${TAG}_bench_bls12381_2p14.json
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt1 -o r -- python bench.py --inflight 1 --steps 8 --warmup 2 --no-cpu-baseline > $O/${TAG}_kt1.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_kt1/r_results.db > $O/${TAG}_kernel_trace_bn254_2p17.txt
grep '"metric"' $O/${TAG}_kt1.log | tail -1 > $O/${TAG}_kernel_trace_bn254_2p17_benchline.json
rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt24 -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/${TAG}_kt24.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_kt24/r_results.db > $O/${TAG}_kernel_trace_bn254_2p17_saturated.txt
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $C -d $O/${TAG}_pmc_$N -o p -- python tools/prof_msm.py 17 4 0 > $O/${TAG}_pmc_$N.log 2>&1
  python tools/pmc_summary.py $O/${TAG}_pmc_$N/p_results.db > $O/${TAG}_pmc_$N.txt
done
rm -rf $O/${TAG}_kt1 $O/${TAG}_kt24 $O/${TAG}_pmc_*/
ls -la $O | tail -20
