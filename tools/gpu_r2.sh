#!/bin/bash
# Round-2 gpurun stages.  usage: gpu_r2.sh "test bench prof traffic dropin kernels"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
WHAT=${1:-test bench}
if [[ $WHAT == *test* ]]; then
  timeout 1200 python -m pytest tests -m gpu -q -s --timeout 600 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  grep -E "passed|failed|FAILED|Error|\[parity\] (SDXL|flash|tiny SD1.5 pipeline on)" $O/pytest_gpu.log | tail -40
fi
if [[ $WHAT == *fullsize* ]]; then
  timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -q -s --timeout 600 > $O/pytest_fullsize.log 2>&1; echo "pytest fullsize rc=$?" | tee -a $O/pytest_fullsize.log
  grep -E "passed|failed|FAILED|Error|\[parity\]" $O/pytest_fullsize.log | tail -30
fi
if [[ $WHAT == *bench* ]]; then
  timeout 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  cat $O/bench.json; grep "^\[bench" $O/bench.err | tail -20
fi
if [[ $WHAT == *prof* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof
  timeout 420 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o sdxl -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference > $O/prof.log 2>&1; echo "prof rc=$?"
  grep '"metric"' $O/prof.log | cut -c1-200
  find $O/prof -name '*kernel_trace*' -size +30M -delete
  cd $R
  python tools/prof_summary.py $(find $O/prof -name '*kernel_stats.csv' | head -1) "r02 sdxl bench (--steps 1 --warmup 1)" > $O/prof_summary.md 2>> $O/prof.log; head -30 $O/prof_summary.md
fi
if [[ $WHAT == *traffic* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/pmc_traffic; mkdir -p $O/pmc_traffic
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_traffic/fetch -o sdxl -- python $R/bench.py --no-graph --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-reference > $O/pmc_traffic/fetch.log 2>&1; echo "pmc fetch rc=$?"
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_traffic/write -o sdxl -- python $R/bench.py --no-graph --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-reference > $O/pmc_traffic/write.log 2>&1; echo "pmc write rc=$?"
  cd $R
  ALGO=$(python -c "import json;print(json.load(open('$O/bench.json'))['roofline']['algorithmic_bytes_per_launch'])" 2>/dev/null)
  python tools/pmc_traffic.py $O/pmc_traffic/fetch $O/pmc_traffic/write $O/r02_sdxl_traffic.md $O/sdxl_traffic.json $ALGO
  find $O/pmc_traffic -name '*kernel_trace*' -delete
  find $O/pmc_traffic -name '*counter_collection.csv' -size +8M -delete
  tail -3 $O/pmc_traffic/fetch.log | cut -c1-300
fi
if [[ $WHAT == *dropin* ]]; then
  timeout 400 python tools/dropin_loop.py > $O/dropin.json 2> $O/dropin.err; echo "dropin rc=$?"
  cat $O/dropin.json; tail -3 $O/dropin.err
fi
if [[ $WHAT == *kernels* ]]; then
  timeout 900 python tools/bench_kernels_r2.py > $O/kernels_r2.log 2>&1; echo "kernels rc=$?"
  grep -E '^\{' $O/kernels_r2.log | cut -c1-260 | tail -80
fi
