#!/bin/bash
# Round-4 gpurun stages.  usage: gpu_r4.sh "attn attntest ..."
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
WHAT=${1:-attn}
if [[ $WHAT == *attntest* ]]; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s --timeout 300 -k "attention" > $O/pytest_attn.log 2>&1; echo "pytest attn rc=$?"
  grep -E "passed|failed|FAILED|Error" $O/pytest_attn.log | tail -20
  grep -E "\[parity\] flash attn (v2|spiked)" $O/pytest_attn.log | tail -30
fi
if [[ $WHAT == *attnbench* ]]; then
  rm -f $O/attn_r4.jsonl
  timeout 600 python tools/bench_attn_r4.py $O/attn_r4.jsonl > $O/attn_r4.log 2>&1; echo "attn bench rc=$?"
  tail -3 $O/attn_r4.log | cut -c1-300
fi
if [[ $WHAT == *benchfast* ]]; then
  timeout 900 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  cut -c1-1200 $O/bench.json; grep "^\[bench" $O/bench.err | tail -20
fi
