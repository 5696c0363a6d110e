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
if [[ $WHAT == *ksweep* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/ksweep; mkdir -p $O/ksweep
  DIFFUSERS_AMD_TUNE=0 timeout 500 rocprofv3 --kernel-trace -f csv -d $O/ksweep -o ks -- python $R/tools/ksweep.py $O/ksweep/manifest.json > $O/ksweep/run.log 2>&1; echo "ksweep rc=$?"
  cd $R
  python tools/ksweep_report.py $O/ksweep/manifest.json $(find $O/ksweep -name '*kernel_trace.csv' | head -1) > $O/ksweep_report.md 2>> $O/ksweep/run.log
  find $O/ksweep -name '*kernel_trace*' -delete
  grep -A40 "K sweep" $O/ksweep_report.md; tail -3 $O/ksweep/run.log
fi
if [[ $WHAT == *fullsize* ]]; then
  timeout 1500 python -m pytest tests/test_full_size_gpu.py -m gpu -q -s --timeout 900 > $O/pytest_fullsize.log 2>&1; echo "pytest fullsize rc=$?" | tee -a $O/pytest_fullsize.log
  grep -E "passed|failed|FAILED|Error|\[parity\]" $O/pytest_fullsize.log | tail -30
fi
if [[ $WHAT == *others* ]]; then
  # the other BASELINE configs under the driver contract; unseen GEMM / conv shapes are tuned live (both kernel families) and saved
  for cfg in sd15 flux ddpm wan; do
    ST=2; [[ $cfg == wan ]] && ST=1
    DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_$cfg.json timeout 1200 python bench.py --config $cfg --steps $ST --warmup 1 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg rc=$?"
    cut -c1-700 $O/bench_$cfg.json; tail -2 $O/bench_$cfg.err
  done
fi
if [[ $WHAT == *pmc* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/pmc_k2; mkdir -p $O/pmc_k2
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -f csv -d $O/pmc_k2/sq -o k2 -- python $R/tools/pmc_k2.py $O/pmc_k2/manifest.json > $O/pmc_k2/sq.log 2>&1; echo "pmc sq rc=$?"
  cd $R
  python tools/pmc_k2_report.py $O/pmc_k2/manifest.json $(find $O/pmc_k2/sq -name '*counter_collection.csv' | head -1) > $O/pmc_k2_report.md 2>> $O/pmc_k2/sq.log
  find $O/pmc_k2 -name '*kernel_trace*' -delete; find $O/pmc_k2 -name '*counter_collection.csv' -size +8M -delete
  cat $O/pmc_k2_report.md; tail -3 $O/pmc_k2/sq.log
fi
if [[ $WHAT == *traffic* ]]; then
  # per-launch HBM-side traffic of the denoising-step kernels (2 eager steps): FETCH and WRITE in separate --pmc passes, stamped
  # with the build fingerprint bench.py compares before it reports roofline.traffic
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/pmc_traffic; mkdir -p $O/pmc_traffic
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_traffic/fetch -o sdxl -- python $R/tools/pmc_one_step.py 2 > $O/pmc_traffic/fetch.log 2>&1; echo "pmc fetch rc=$?"
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_traffic/write -o sdxl -- python $R/tools/pmc_one_step.py 2 > $O/pmc_traffic/write.log 2>&1; echo "pmc write rc=$?"
  cd $R
  ALGO=$(python -c "import json;print(json.load(open('$O/bench.json'))['roofline']['algorithmic_bytes_per_launch'])" 2>/dev/null)
  python tools/pmc_traffic.py $O/pmc_traffic/fetch $O/pmc_traffic/write $O/r03_sdxl_traffic.md $O/sdxl_traffic.json $ALGO
  find $O/pmc_traffic -name '*kernel_trace*' -delete
  find $O/pmc_traffic -name '*counter_collection.csv' -size +8M -delete
  tail -4 $O/pmc_traffic/fetch.log | cut -c1-200
fi
if [[ $WHAT == *reothers* ]]; then
  # re-measure the variant choice of every shape of the other BASELINE configs with the current kernels (empty table), merge
  # the results over the shipped table, then run each config's driver-contract line on the merged table
  python - <<PYEOF
import json, shutil
shutil.copy("$R/diffusers_amd/tuned/gfx950.json", "$O/tuned_all.json")
PYEOF
  for cfg in sd15 flux ddpm wan; do
    rm -f $O/tuned_$cfg.json
    DIFFUSERS_AMD_TUNE_DB=$O/none.json DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_$cfg.json timeout 1500 python bench.py --config $cfg --steps 1 --warmup 1 > $O/retune_$cfg.json 2> $O/retune_$cfg.err; echo "retune $cfg rc=$?"
    python - <<PYEOF
import json
a = json.load(open("$O/tuned_all.json"))
b = json.load(open("$O/tuned_$cfg.json"))
a["entries"].update(b["entries"]); a["format"], a["tiles"] = b["format"], b["tiles"]
json.dump(a, open("$O/tuned_all.json", "w"), indent=0)
print("$cfg:", len(b["entries"]), "shapes tuned; table now", len(a["entries"]))
PYEOF
  done
  for cfg in sd15 flux ddpm wan; do
    ST=2; [[ $cfg == wan ]] && ST=1
    DIFFUSERS_AMD_TUNE_DB=$O/tuned_all.json timeout 1200 python bench.py --config $cfg --steps $ST --warmup 1 > $O/bench_$cfg.json 2> $O/bench_$cfg.err; echo "$cfg rc=$?"
    cut -c1-330 $O/bench_$cfg.json
  done
fi
