cd $GRAFT_REPO_ROOT
timeout 300 python tools/bench_attn_r3.py 2>&1 | tail -5 | cut -c1-500
