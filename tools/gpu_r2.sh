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
if [[ $WHAT == *retune* ]]; then
  # re-measure the variant choice (tile, ring depth, split-K, paired launches) of every SDXL shape with the current kernels
  rm -f $O/tuned_sdxl_r2.json
  DIFFUSERS_AMD_TUNE_DB=$O/none.json DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_sdxl_r2.json timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference > $O/retune.json 2> $O/retune.err; echo "retune rc=$?"
  cut -c1-200 $O/retune.json
  python - <<PYEOF
import json
a = json.load(open("$R/diffusers_amd/tuned/gfx950.json"))
b = json.load(open("$O/tuned_sdxl_r2.json"))
ch = sum(1 for k, v in b["entries"].items() if k in a["entries"] and a["entries"][k][:2] != v[:2])
sp = sum(1 for v in b["entries"].values() if len(v) > 3)
pr = sum(1 for k in b["entries"] if k.startswith("pair:"))
a["entries"].update(b["entries"])
a["format"] = b["format"]
json.dump(a, open("$O/tuned_merged_r2.json", "w"), indent=0)
print("retuned", len(b["entries"]), "shapes;", ch, "changed variant;", sp, "use split-K;", pr, "paired; table now", len(a["entries"]))
PYEOF
  export DIFFUSERS_AMD_TUNE_DB=$O/tuned_merged_r2.json
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
if [[ $WHAT == *onestep* ]]; then
  # per-launch HBM-side traffic from a SMALL eager workload (a few denoising steps): FETCH and WRITE in separate passes
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/pmc_traffic; mkdir -p $O/pmc_traffic
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_traffic/fetch -o sdxl -- python $R/tools/pmc_one_step.py 2 > $O/pmc_traffic/fetch.log 2>&1; echo "pmc fetch rc=$?"
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_traffic/write -o sdxl -- python $R/tools/pmc_one_step.py 2 > $O/pmc_traffic/write.log 2>&1; echo "pmc write rc=$?"
  cd $R
  ALGO=$(python -c "import json;print(json.load(open('$O/bench.json'))['roofline']['algorithmic_bytes_per_launch'])" 2>/dev/null)
  python tools/pmc_traffic.py $O/pmc_traffic/fetch $O/pmc_traffic/write $O/r02_sdxl_traffic.md $O/sdxl_traffic.json $ALGO
  find $O/pmc_traffic -name '*kernel_trace*' -delete
  find $O/pmc_traffic -name '*counter_collection.csv' -size +8M -delete
  tail -4 $O/pmc_traffic/fetch.log | cut -c1-200
fi
if [[ $WHAT == *dropin* ]]; then
  timeout 400 python tools/dropin_loop.py > $O/dropin.json 2> $O/dropin.err; echo "dropin rc=$?"
  cat $O/dropin.json; tail -3 $O/dropin.err
fi
if [[ $WHAT == *kernels* ]]; then
  timeout 900 python tools/bench_kernels_r2.py > $O/kernels_r2.log 2>&1; echo "kernels rc=$?"
  grep -E '^\{' $O/kernels_r2.log | grep -v "splitk.var" | cut -c1-300 | tail -90
  grep -vE '^\{' $O/kernels_r2.log | tail -15
fi
if [[ $WHAT == *lnfold* ]]; then
  timeout 300 python tools/bench_lnfold.py > $O/lnfold.log 2>&1; echo "lnfold rc=$?"; grep -E '^\{' $O/lnfold.log | cut -c1-220; grep -vE '^\{' $O/lnfold.log | tail -5
  timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s --timeout 200 -k "layernorm_fold or conv_split_k" > $O/pytest_ln.log 2>&1; echo "pytest ln rc=$?"; grep -E "passed|failed|FAILED|Error" $O/pytest_ln.log | tail -5
fi
if [[ $WHAT == *textenc* ]]; then
  timeout 600 python -m pytest tests/test_text_encoders.py tests/test_kernels_gpu.py -m gpu -q -s --timeout 300 -k "text_encoder or masked_flash or clip or t5 or layernorm_fold" > $O/pytest_textenc.log 2>&1; echo "pytest textenc rc=$?" | tee -a $O/pytest_textenc.log
  grep -E "passed|failed|FAILED|Error|\[parity\] (CLIP|T5|UMT5|masked)" $O/pytest_textenc.log | tail -30
fi
if [[ $WHAT == *newtests* ]]; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s --timeout 300 -k "pair_launch or split_k or sampler_round2 or flowmatch_fp32 or flash_attention or attention or torch_library" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/pytest_new.log
  grep -E "passed|failed|FAILED|Error|split-K" $O/pytest_new.log | tail -30
fi
if [[ $WHAT == *r2b* ]]; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s --timeout 300 -k "bit_identical or query_block or tuner or flash_attention" > $O/pytest_r2b.log 2>&1; echo "pytest r2b rc=$?" | tee -a $O/pytest_r2b.log
  grep -E "passed|failed|FAILED|Error" $O/pytest_r2b.log | tail -12
  timeout 600 python tools/bench_kernels_r2b.py > $O/kernels_r2b.log 2>&1; echo "kernels r2b rc=$?"
  grep -E '^\{' $O/kernels_r2b.log | cut -c1-400; grep -vE '^\{' $O/kernels_r2b.log | tail -8
fi
if [[ $WHAT == *ksweep* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/ksweep; mkdir -p $O/ksweep
  DIFFUSERS_AMD_TUNE=0 timeout 400 rocprofv3 --kernel-trace -f csv -d $O/ksweep -o ks -- python $R/tools/ksweep.py $O/ksweep/manifest.json > $O/ksweep/run.log 2>&1; echo "ksweep rc=$?"
  cd $R
  python tools/ksweep_report.py $O/ksweep/manifest.json $(find $O/ksweep -name '*kernel_trace.csv' | head -1) > $O/ksweep_report.md 2>> $O/ksweep/run.log
  find $O/ksweep -name '*kernel_trace*' -delete
  grep -A40 "K sweep" $O/ksweep_report.md; tail -3 $O/ksweep/run.log
fi
if [[ $WHAT == *twostream* ]]; then
  DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_batch1.json timeout 800 python tools/bench_two_stream.py > $O/two_stream.log 2>&1; echo "twostream rc=$?"
  grep -E '^\{' $O/two_stream.log; grep -vE '^\{' $O/two_stream.log | tail -8
fi
if [[ $WHAT == *wanvae* ]]; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -m gpu -q -s --timeout 300 -k "k_valid or wan_vae or decodes_video or all_variants" > $O/pytest_wan.log 2>&1; echo "pytest wan rc=$?" | tee -a $O/pytest_wan.log
  grep -E "passed|failed|FAILED|Error|\[parity\]" $O/pytest_wan.log | tail -12
  # every shape of the decoder re-tuned with the current kernels (empty table), with and without k_valid
  DIFFUSERS_AMD_TUNE_DB=$O/none.json DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_wan_vae.json timeout 900 python tools/bench_wan_vae.py > $O/wan_vae.log 2>&1; echo "wanvae rc=$?"
  grep -E '^\{|bench_wan_vae' $O/wan_vae.log | tail -8; grep -vE '^\{|bench_wan_vae' $O/wan_vae.log | tail -5
fi
if [[ $WHAT == *others* ]]; then
  # the other BASELINE configs, every GEMM / conv shape re-tuned with the current kernels (empty table) and saved
  for cfg in flux sd15 ddpm wan; do
    DIFFUSERS_AMD_TUNE_DB=$O/none.json DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_$cfg.json timeout 900 python tools/bench_$cfg.py > $O/$cfg.log 2>&1; echo "$cfg rc=$?"
    grep -E '^\{' $O/$cfg.log | cut -c1-400; grep -vE '^\{' $O/$cfg.log | grep -v amdgpu.ids | tail -2
  done
fi
if [[ $WHAT == *fluxprof* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof_flux
  timeout 420 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_flux -o flux -- python $R/tools/bench_flux.py > $O/prof_flux.log 2>&1; echo "fluxprof rc=$?"
  find $O/prof_flux -name '*kernel_trace*' -size +30M -delete
  cd $R
  python tools/prof_summary.py $(find $O/prof_flux -name '*kernel_stats.csv' | head -1) "flux-schnell (tools/bench_flux.py)" > $O/prof_flux_summary.md 2>> $O/prof_flux.log; head -28 $O/prof_flux_summary.md
fi
if [[ $WHAT == *pvdelay* ]]; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_full_size_gpu.py -m gpu -q -s --timeout 300 -k "flash_attention or attention" > $O/pytest_pv.log 2>&1; echo "pytest pv rc=$?" | tee -a $O/pytest_pv.log
  grep -E "passed|failed|FAILED|Error" $O/pytest_pv.log | tail -8
  timeout 600 python tools/bench_kernels_r2b.py attn > $O/attn_pv.log 2>&1; echo "attn pv rc=$?"
  grep -E '^\{' $O/attn_pv.log | cut -c1-300; grep -vE '^\{' $O/attn_pv.log | tail -5
fi
