cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x --timeout 300 -k "group or gn or norm" 2>&1 | tail -3
bash tools/gpu_r3.sh "benchfast" | tail -3 | cut -c1-200
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json"))
k={x["name"]:x for x in d["roofline"]["kernels"]}
print(d["value"], {n:(round(v["ms"],2), v.get("launches"), round(v["frac"],3)) for n,v in k.items()})
PY
