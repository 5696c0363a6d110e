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
if [[ $WHAT == *others* ]]; then
  for cfg in flux wan sd15; do
    ST=2; [[ $cfg == wan ]] && ST=1
    timeout 900 python bench.py --config $cfg --steps $ST --warmup 1 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg rc=$?"
    cut -c1-400 $O/bench_$cfg.json; tail -2 $O/bench_$cfg.err | cut -c1-300
  done
fi
if [[ $WHAT == *lnfold* ]]; then
  timeout 600 python -m pytest tests/test_gemm_k2_gpu.py tests/test_kernels_gpu.py -m gpu -q -s --timeout 300 -k "layernorm_fold" > $O/pytest_lnfold.log 2>&1; echo "pytest lnfold rc=$?"
  grep -E "passed|failed|FAILED|Error|assert" $O/pytest_lnfold.log | tail -20
  rm -f $O/lnfold_r4.jsonl
  timeout 300 python tools/bench_lnfold_r4.py $O/lnfold_r4.jsonl > $O/lnfold_r4.log 2>&1; echo "lnfold bench rc=$?"
  cat $O/lnfold_r4.jsonl | cut -c1-600
fi
if [[ $WHAT == *norms* ]]; then
  rm -f $O/norms_r4.jsonl
  timeout 120 python tools/bench_norms_r4.py $O/norms_r4.jsonl > $O/norms_r4.log 2>&1; echo "norms default rc=$?"
  for kn in "DA_GN_THREADS=512" "DA_GN_MINPIX=16" "DA_GN_MINPIX=4" "DA_GN_MAXBLK=1024 DA_GN_CAP=4096" "DA_GN_THREADS=512 DA_GN_MINPIX=16" "DA_GN_MAXBLK=256"; do
    env $kn timeout 120 python tools/bench_norms_r4.py $O/norms_r4.jsonl >> $O/norms_r4.log 2>&1
  done
  grep "sum over\|layernorm" $O/norms_r4.jsonl | cut -c1-200
fi
if [[ $WHAT == *foldab* ]]; then
  for mode in 0 2 1 0 2; do
    DIFFUSERS_AMD_LN_FOLD=$mode timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline > $O/bench_fold$mode.json 2> $O/bench_fold$mode.err; echo "fold mode $mode rc=$? $(cut -c1-140 $O/bench_fold$mode.json | grep -o '"value": [0-9.]*')"
  done
fi
if [[ $WHAT == *normsdef* ]]; then
  rm -f $O/norms_r4b.jsonl
  timeout 120 python tools/bench_norms_r4.py $O/norms_r4b.jsonl > $O/norms_r4b.log 2>&1; echo "norms rc=$?"
  grep "sum over\|layernorm" $O/norms_r4b.jsonl | cut -c1-200
fi
if [[ $WHAT == *benchfull* ]]; then
  timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
  cut -c1-300 $O/bench_full.json; grep "^\[bench" $O/bench_full.err | tail -40
fi
if [[ $WHAT == *fulltest* ]]; then
  timeout 2400 python -m pytest tests -m gpu -q -s --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -20
  grep -E "\[parity\] (SDXL|FLUX|Wan|SD1.5|full)|\[drop-in\]" $O/pytest_gpu.log | tail -40
fi
