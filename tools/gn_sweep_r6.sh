#!/bin/bash
# Round 6: GroupNorm plan knobs after the prologue fix (tools/bench_norms_r4.py: chained us per launch at the 14 GroupNorm shapes of an SDXL step)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" timeout 200 python tools/bench_norms_r4.py 2>/dev/null | grep -E "sum over|16384.*320\+0|16384.*640\+0|4096.*1280\+0|4096.*640\+0\"" | cut -c1-150; }
run DA_X=0
run DA_GN_MAXBLK=192
run DA_GN_MAXBLK=128
run DA_GN_MAXBLK=96
run DA_GN_MAXBLK=64
run DA_GN_MAXBLK=32
run DA_GN_MAXBLK=128 DA_GN_THREADS=1024
run DA_GN_MAXBLK=64 DA_GN_THREADS=1024
run DA_X=0
