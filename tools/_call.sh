cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline > $O/bench_A.json 2> $O/bench_A.err; echo "bench A (shipped table) rc=$?"; cut -c1-260 $O/bench_A.json
bash tools/gpu_r3.sh "retune benchfast"
cp $O/bench.json $O/bench_B.json
unset DIFFUSERS_AMD_TUNE_DB
bash tools/gpu_r3.sh "fullsize"
timeout 300 python tools/bench_attn_r3.py > $O/bench_attn_r3.txt 2>&1; echo "attn rc=$?"; tail -30 $O/bench_attn_r3.txt
bash tools/gpu_r3.sh "others"
