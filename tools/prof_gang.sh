# kernel traces of the loaded prover with and without gangs (measurement aid): bash tools/prof_gang.sh <curve> <log_n>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out/r6e
for g in 1 2; do
  rm -rf /tmp/prof_g$g
  APK_GANG=$g rocprofv3 --kernel-trace -d /tmp/prof_g$g -o p -- python $R/bench.py --curve $1 --log-n $2 --inflight 32 --steps 6 --warmup 2 --no-pmc --no-cpu-baseline --no-host-inputs --no-oracle-check > $R/gpurun_out/r6e/bench_g$g.json 2> $R/gpurun_out/r6e/err_g$g.txt
  db=$(find /tmp/prof_g$g -name "*_results.db" | head -1)
  python $R/tools/rocprof_summary.py $db > $R/gpurun_out/r6e/trace_$1_$2_g$g.txt
  python -c "
import json;d=json.load(open('$R/gpurun_out/r6e/bench_g$g.json'));print('gang $g:',d['value'],d['paths_under_load']['proofs'])"
  python $R/tools/stream_timeline.py $db 150 150 > $R/gpurun_out/r6e/streams_$1_$2_g$g.txt
  head -24 $R/gpurun_out/r6e/streams_$1_$2_g$g.txt
done
