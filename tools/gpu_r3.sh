#!/bin/bash
# Round-3 gpurun stages.  usage: gpu_r3.sh "retune bench test prof"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
WHAT=${1:-bench}
if [[ $WHAT == *retune* ]]; then
  # re-measure the variant choice (both kernel families, paired launches) of every SDXL shape with the current kernels
  rm -f $O/tuned_sdxl_r3.json
  DIFFUSERS_AMD_TUNE_DB=$O/none.json DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_sdxl_r3.json timeout 700 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference > $O/retune.json 2> $O/retune.err; echo "retune rc=$?"
  cut -c1-200 $O/retune.json
  python - <<PYEOF
import json
a = json.load(open("$R/diffusers_amd/tuned/gfx950.json"))
b = json.load(open("$O/tuned_sdxl_r3.json"))
ch = sum(1 for k, v in b["entries"].items() if k in a["entries"] and a["entries"][k][:2] != v[:2])
k2 = sum(1 for v in b["entries"].values() if v[0] >= 9)
pr = sum(1 for k in b["entries"] if k.startswith("pair:"))
a["entries"].update(b["entries"])
a["format"], a["tiles"] = b["format"], b["tiles"]
json.dump(a, open("$O/tuned_merged_r3.json", "w"), indent=0)
print("retuned", len(b["entries"]), "shapes;", ch, "changed variant;", k2, "use the K2 family;", pr, "paired; table now", len(a["entries"]))
PYEOF
  export DIFFUSERS_AMD_TUNE_DB=$O/tuned_merged_r3.json
fi
if [[ $WHAT == *bench* ]]; then
  EXTRA=""
  [[ $WHAT == *benchfast* ]] && EXTRA="--no-reference --no-cpu-baseline"
  timeout 1200 python bench.py --steps 3 --warmup 1 $EXTRA > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  cut -c1-1500 $O/bench.json; grep "^\[bench" $O/bench.err | tail -20
fi
if [[ $WHAT == *test* ]]; then
  timeout 1500 python -m pytest tests -m gpu -q -s --timeout 600 -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  grep -E "passed|failed|FAILED|Error|\[parity\] (SDXL|flash|tiny SD1.5 pipeline on)" $O/pytest_gpu.log | tail -40
fi
if [[ $WHAT == *prof* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof
  timeout 420 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o sdxl -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference > $O/prof.log 2>&1; echo "prof rc=$?"
  grep '"metric"' $O/prof.log | cut -c1-200
  find $O/prof -name '*kernel_trace*' -size +30M -delete
  cd $R
  python tools/prof_summary.py $(find $O/prof -name '*kernel_stats.csv' | head -1) "r03 sdxl bench (--steps 1 --warmup 1)" > $O/prof_summary.md 2>> $O/prof.log; head -40 $O/prof_summary.md
fi
