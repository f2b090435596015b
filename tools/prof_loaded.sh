# kernel trace of the loaded prover (measurement aid): bash tools/prof_loaded.sh <curve> <log_n> <inflight> <tag> [ENV=VAL ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
curve=$1; logn=$2; inflight=$3; tag=$4; shift 4
mkdir -p $R/gpurun_out/prof_$tag
rm -rf /tmp/prof_$tag
env "$@" rocprofv3 --kernel-trace -d /tmp/prof_$tag -o p -- python $R/bench.py --curve $curve --log-n $logn --inflight $inflight --steps 10 --warmup 2 --no-pmc --no-cpu-baseline --no-host-inputs --no-oracle-check > $R/gpurun_out/prof_$tag/bench.json 2> $R/gpurun_out/prof_$tag/err.txt
db=$(find /tmp/prof_$tag -name "*_results.db" | head -1)
python $R/tools/rocprof_summary.py $db > $R/gpurun_out/prof_$tag/trace.txt
python $R/tools/stream_timeline.py $db 100 400 > $R/gpurun_out/prof_$tag/streams.txt
python -c "
import json;d=json.load(open('$R/gpurun_out/prof_$tag/bench.json'));print('$tag:',d['value'],d['paths_under_load'])"
head -30 $R/gpurun_out/prof_$tag/trace.txt | cut -c1-125
head -4 $R/gpurun_out/prof_$tag/streams.txt
