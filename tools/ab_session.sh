cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
echo "== previous / new partition scan (value, latency, acc, sat MSM, ntt)"
bash tools/ab_libs.sh 2 "--steps 30" algoplonk_amd/libapk_prev.so algoplonk_amd/libapk.so
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/s6_tl -o r -- python tools/prof_msm.py 17 0 4 > gpurun_out/s6_tl.log 2>&1
python tools/timeline.py gpurun_out/s6_tl/r_results.db 3.9 > gpurun_out/s6_timeline_final.txt
rm -rf gpurun_out/s6_tl
grep "msm_part" gpurun_out/s6_timeline_final.txt | head -4; tail -1 gpurun_out/s6_timeline_final.txt
