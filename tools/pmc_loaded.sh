# VALU instructions per kernel of the LOADED prover (measurement aid): bash tools/pmc_loaded.sh <curve> <log_n> <callers> <rounds> <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
rm -rf /tmp/pmcl_$5
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES -d /tmp/pmcl_$5 -o p -- python $R/tools/prof_loaded_proofs.py $1 $2 $3 $4 > $R/gpurun_out/pmc_loaded_$5.log 2>&1
python $R/tools/pmc_summary.py $(find /tmp/pmcl_$5 -name "p_results.db" | head -1) > $R/gpurun_out/pmc_loaded_$5.txt
tail -2 $R/gpurun_out/pmc_loaded_$5.log; cut -c1-150 $R/gpurun_out/pmc_loaded_$5.txt | head -45
