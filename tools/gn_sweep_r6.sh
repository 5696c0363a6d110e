cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 200 python tools/bench_norms_r4.py 2>/dev/null | grep -E "sum over|16384.*320\+0|4096.*640\+0\"" | cut -c1-200; }
run DA_X=0
run DA_GN_MAXBLK=256
run DA_GN_MAXBLK=128
run DA_GN_MAXBLK=1024 DA_GN_CAP=4096
run DA_GN_MINPIX=4
run DA_GN_MINPIX=32
run DA_GN_MINPIX=64
run DA_GN_THREADS=256
run DA_GN_THREADS=1024
run DA_GN_THREADS=256 DA_GN_MINPIX=32
run DA_GN_THREADS=512 DA_GN_MINPIX=32 DA_GN_MAXBLK=256
