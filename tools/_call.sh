cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline > $O/bench_A.json 2> $O/bench_A.err; echo "bench A (shipped table) rc=$?"; cut -c1-160 $O/bench_A.json
bash tools/gpu_r3.sh "retune benchfast" | tail -4 | cut -c1-200
cp $O/bench.json $O/bench_B.json
python - <<'PY'
import json
for n in ("A","B"):
    d=json.load(open(f"gpurun_out/bench_{n}.json"))
    k={x["name"]:x for x in d["roofline"]["kernels"]}
    print(n, d["value"], "igemm ms", round(k["igemm"]["ms"],2), "frac", round(k["igemm"]["frac"],4), "attn ms", round(k["attention"]["ms"],2), "ln ms", round(k["layernorm"]["ms"],2), "vae ms", round(k["vae_decode"]["ms"],2), "tuned_live", d["config"].get("tuned_live"))
PY
unset DIFFUSERS_AMD_TUNE_DB
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -4
