cd $GRAFT_REPO_ROOT
bash tools/gpu_r3.sh "reothers"
