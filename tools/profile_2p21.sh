#!/bin/bash
# BLS12-381 2^21 (+1 BSB22 commitment): kernel trace and PMC passes over two proofs, one at a time.  Run through gpurun from the
# repo root:  tools/profile_2p21.sh TAG  -> gpurun_out/TAG_*  (copy what is to be judged into profiles/)
TAG=${1:-rXX}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
ARGS="--curve bls12_381 --log-n 21 --bsb22 1 --inflight 1 --steps 2 --warmup 1 --no-pmc --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt21 -o r -- python bench.py $ARGS > $O/${TAG}_kt21.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_kt21/r_results.db > $O/${TAG}_kernel_trace_bls12381_2p21.txt
grep '"metric"' $O/${TAG}_kt21.log | tail -1 > $O/${TAG}_kernel_trace_bls12381_2p21_benchline.json
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  N=$(echo $C | tr ' ' '_')
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $O/${TAG}_pmc21_$N -o p -- python bench.py $ARGS > $O/${TAG}_pmc21_$N.log 2>&1
  python tools/pmc_summary.py $O/${TAG}_pmc21_$N/p_results.db > $O/${TAG}_pmc_bls12381_2p21_$N.txt
done
rm -rf $O/${TAG}_kt21 $O/${TAG}_pmc21_*/
ls -la $O | grep ${TAG}
