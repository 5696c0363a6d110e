cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gemm_k2_gpu.py tests/test_kernels_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -5
timeout 400 python tools/bench_k2.py geglu > $O/bench_k2_sw.jsonl 2> $O/bench_k2_sw.err; echo "bench_k2 rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/bench_k2_sw.jsonl'):
    r=json.loads(l)
    vs={k:v for k,v in r.items() if isinstance(v,list) and '/' in k}
    best=sorted((v[2],k) for k,v in vs.items())[:4]
    print(r['name'], 'table', r['table'], 'best chain', best, {k:v for k,v in vs.items() if '320' in k})
PY
