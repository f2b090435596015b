#!/bin/bash
# Collects what profiles/ keeps for a round (run through gpurun from the repo root): bench lines, kernel traces, PMC passes.
# usage: tools/profile_round.sh TAG     -> gpurun_out/TAG_*
TAG=${1:-rXX}
O=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 600 python bench.py > $O/${TAG}_bench_bn254_2p17.log 2>&1; tail -1 $O/${TAG}_bench_bn254_2p17.log > $O/${TAG}_bench_bn254_2p17.json
timeout 500 python bench.py --curve bls12_381 --log-n 14 > $O/${TAG}_bench_bls12381_2p14.log 2>&1; tail -1 $O/${TAG}_bench_bls12381_2p14.log > $O/${TAG}_bench_bls12381_2p14.json
timeout 3000 python bench.py --curve bls12_381 --log-n 21 --bsb22 1 --inflight 8 --steps 4 --warmup 1 $BLS21_FLAGS > $O/${TAG}_bench_bls12381_2p21_bsb22.log 2>&1; tail -1 $O/${TAG}_bench_bls12381_2p21_bsb22.log > $O/${TAG}_bench_bls12381_2p21_bsb22.json
timeout 500 python bench.py --curve bls12_381 --log-n 14 --inflight 64 > $O/${TAG}_bench_bls12381_2p14_64callers.log 2>&1; tail -1 $O/${TAG}_bench_bls12381_2p14_64callers.log > $O/${TAG}_bench_bls12381_2p14_64callers.json
timeout 200 python bench.py --mode msm-sharded --steps 50 > $O/${TAG}_bench_msm_sharded.log 2>&1; tail -1 $O/${TAG}_bench_msm_sharded.log > $O/${TAG}_bench_msm_sharded.json
timeout 200 python bench.py --mode prove-split --curve bls12_381 --log-n 21 --steps 5 --warmup 1 > $O/${TAG}_bench_prove_split_2p21.log 2>&1; tail -1 $O/${TAG}_bench_prove_split_2p21.log > $O/${TAG}_bench_prove_split_2p21.json
# kernel traces: one proof at a time (sequential) and the bench's 32 callers (saturated)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt1 -o r -- python bench.py --inflight 1 --steps 8 --warmup 2 --no-cpu-baseline --no-pmc --no-oracle-check > $O/${TAG}_kt1.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_kt1/r_results.db > $O/${TAG}_kernel_trace_bn254_2p17.txt
grep '"metric"' $O/${TAG}_kt1.log | tail -1 > $O/${TAG}_kernel_trace_bn254_2p17_benchline.json
# ... and twelve lone proofs and nothing else (tools/prof_msm.py): the population bench.py's HIP-event statistics time - the average
# msm_accumulate_kernel duration of THIS table is the one to hold against the line's roofline.avg_launch_ms
APK_PROF_SLOTS=32 APK_PROF_STATS=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_ktlone -o r -- python tools/prof_msm.py 17 0 48 > $O/${TAG}_ktlone.log 2>&1
(grep "HIP events" $O/${TAG}_ktlone.log; python tools/rocprof_summary.py $O/${TAG}_ktlone/r_results.db) > $O/${TAG}_kernel_trace_bn254_2p17_lone_proofs.txt
rm -rf $O/${TAG}_ktlone
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_kt24 -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-oracle-check --no-host-inputs > $O/${TAG}_kt24.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_kt24/r_results.db > $O/${TAG}_kernel_trace_bn254_2p17_saturated.txt
python tools/stream_timeline.py $O/${TAG}_kt24/r_results.db 150 60 > $O/${TAG}_streams_bn254_2p17_saturated.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_ktbls64 -o r -- python bench.py --curve bls12_381 --log-n 14 --inflight 64 --steps 6 --warmup 1 --no-cpu-baseline --no-pmc --no-host-inputs --no-oracle-check > $O/${TAG}_ktbls64.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_ktbls64/r_results.db > $O/${TAG}_kernel_trace_bls12381_2p14_gangs.txt
python tools/stream_timeline.py $O/${TAG}_ktbls64/r_results.db 100 60 > $O/${TAG}_streams_bls12381_2p14_gangs.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $O/${TAG}_ktbls -o r -- python bench.py --curve bls12_381 --log-n 14 --inflight 1 --steps 8 --warmup 2 --no-cpu-baseline --no-pmc --no-oracle-check > $O/${TAG}_ktbls.log 2>&1
python tools/rocprof_summary.py $O/${TAG}_ktbls/r_results.db > $O/${TAG}_kernel_trace_bls12381_2p14.txt
# PMC passes (one counter group per pass, kernel trace only: no other trace domains).  Every pass leaves its per-kernel table as
# text AND as JSON (with prof_msm.py's facts about the run); tools/pmc_accumulate.py writes the summary FROM those.
pmc_pass() {   # config-name  counters  prof_msm args...   (extra environment through PMC_ENV)
  local CFG=$1 C=$2; shift 2
  local N=$(echo $C | tr ' ' '_')
  env $PMC_ENV APK_PROF_FACTS=$O/${TAG}_facts_${CFG}.json timeout 400 rocprofv3 --kernel-trace --pmc $C -d $O/${TAG}_pmc_${CFG}_$N -o p -- python tools/prof_msm.py "$@" > $O/${TAG}_pmc_${CFG}_$N.log 2>&1
  python tools/pmc_summary.py $O/${TAG}_pmc_${CFG}_$N/p_results.db --json $O/${TAG}_pmc_${CFG}_$N.json $O/${TAG}_facts_${CFG}.json > $O/${TAG}_pmc_${CFG}_$N.txt
  rm -rf $O/${TAG}_pmc_${CFG}_$N/
}
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"; do
  PMC_ENV= pmc_pass bn254_2p17 "$C" 17 4 2
done
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  PMC_ENV= pmc_pass bls12381_2p14 "$C" 14 4 2 bls12_381
done
# the Infinity Cache question: the same passes with a table LARGER than the 256 MiB cache (2^19 bases x 16
# windows x 64 B = 537 MB) - what FETCH_SIZE reads there is HBM traffic, whatever it was at 2^17 (134-143 MB table)
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  PMC_ENV=APK_MSM_WINDOW=16 pmc_pass bn254_2p19_c16 "$C" 19 4 0
  PMC_ENV=APK_MSM_WINDOW=16 pmc_pass bn254_2p17_c16 "$C" 17 4 0
done
python tools/pmc_accumulate.py $O $TAG $O/${TAG}_pmc_msm_accumulate.json > $O/${TAG}_pmc_accumulate.log 2>&1
# one MSM at a time over the sizes and the window widths (round 5: 18..20 bits from 2^20 bases)
timeout 900 python tools/msm_size_sweep.py 11 24 bn254 $O/${TAG}_msm_size_sweep.json > $O/${TAG}_msm_size_sweep.log 2>&1
timeout 600 python tools/msm_size_sweep.py 19 22 bn254 $O/${TAG}_msm_window_sweep_bn254.json 16,18,19,20 > $O/${TAG}_msm_window_sweep_bn254.log 2>&1
timeout 600 python tools/msm_size_sweep.py 19 21 bls12_381 $O/${TAG}_msm_window_sweep_bls12381.json 16,18,19,20 > $O/${TAG}_msm_window_sweep_bls12381.log 2>&1
# kernel timeline of one lone proof (start, duration, idle gap per launch)
timeout 300 rocprofv3 --kernel-trace -d /tmp/${TAG}_tl17 -o r -- python tools/prof_msm.py 17 0 4 > /dev/null 2>&1
python tools/timeline.py $(find /tmp/${TAG}_tl17 -name r_results.db | head -1) 3.35 > $O/${TAG}_timeline_bn254_2p17.txt
timeout 300 rocprofv3 --kernel-trace -d /tmp/${TAG}_tl14 -o r -- python tools/prof_msm.py 14 0 4 bls12_381 > /dev/null 2>&1
python tools/timeline.py $(find /tmp/${TAG}_tl14 -name r_results.db | head -1) 2.85 > $O/${TAG}_timeline_bls12381_2p14.txt
rm -rf $O/${TAG}_kt1 $O/${TAG}_kt24 $O/${TAG}_ktbls $O/${TAG}_ktbls64 $O/${TAG}_facts_*.json
ls -la $O | grep ${TAG} | tail -40
