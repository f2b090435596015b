cd $GRAFT_REPO_ROOT
timeout 400 python tools/debug/dbg_comm.py 2>&1 | tail -120
