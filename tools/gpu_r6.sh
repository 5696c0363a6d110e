#!/bin/bash
# Round-6 gpurun stages.  usage: gpu_r6.sh "stage stage ..." -- stages run in FILE order, selected by substring.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
WHAT=${1:-attnsplit}
if [[ $WHAT == *retunesplit* ]]; then
  # the shipped table's split-K entries (small-M, deep-K convs of the SD1.5 / DDPM U-Nets) tuned again with split factors up to 24 competing
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 600 -k "split_k" > $O/pytest_splitk.log 2>&1; echo "pytest splitk rc=$?"
  tail -3 $O/pytest_splitk.log
  python - <<PYEOF
import json
R, O = "$R", "$O"
d = json.load(open(f"{R}/diffusers_amd/tuned/gfx950.json"))
drop = [k for k, v in d["entries"].items() if len(v) > 3 and v[3] > 1]
for k in drop:
    del d["entries"][k]
json.dump(d, open(f"{O}/table_pruned.json", "w"))
print("entries to tune again:", len(drop))
PYEOF
  cp $O/table_pruned.json $O/table_retuned.json
  for cfg in sd15 ddpm; do
    DIFFUSERS_AMD_SPLITK=1 DIFFUSERS_AMD_GEMM_FAMILY=all DIFFUSERS_AMD_TUNE_DB=$O/table_retuned.json DIFFUSERS_AMD_TUNE_SAVE=$O/table_retuned.json timeout 900 python bench.py --config $cfg --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/retune_$cfg.json 2> $O/retune_$cfg.err; echo "retune $cfg rc=$? $(grep -o '"value": [0-9.]*' $O/retune_$cfg.json | head -1) $(grep -o '"tuned_live": [0-9]*' $O/retune_$cfg.json)"
  done
  python - <<PYEOF
import json
a = json.load(open("$R/diffusers_amd/tuned/gfx950.json"))["entries"]
b = json.load(open("$O/table_retuned.json"))
T = b["tiles"]
n = 0
for k, v in sorted(b["entries"].items()):
    o = a.get(k)
    if o is None or o[:2] != v[:2] or o[3:] != v[3:]:
        n += 1
        print(f"  {k:58s} {T[o[0]] if o else '-':11s} st{o[1] if o else '-'} split {o[3] if o and len(o) > 3 else 1} {o[2] if o else 0:7.1f}us -> {T[v[0]]:11s} st{v[1]} split {v[3] if len(v) > 3 else 1} {v[2]:7.1f}us")
print("changed:", n)
PYEOF
  for rep in 1 2; do
    for tb in $R/diffusers_amd/tuned/gfx950.json $O/table_retuned.json; do
      for cfg in sd15 ddpm; do
        DIFFUSERS_AMD_TUNE_DB=$tb timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/ab_$cfg.json 2> $O/ab_$cfg.err; echo "$(basename $tb) $cfg rc=$? $(grep -o '"value": [0-9.]*' $O/ab_$cfg.json | head -1) $(grep -o '"tuned_live": [0-9]*' $O/ab_$cfg.json) $(grep -o '"psnr[a-z_]*": [0-9.]*' $O/ab_$cfg.json | head -2 | tr '\n' ' ')"
      done
    done
  done
fi
if [[ $WHAT == *retunesdxl* ]]; then
  # third session: the SDXL U-Net entries (convs and plain nn.Linear; not the k3 GEGLU tiles, pair / qkv launches or the VAE) tuned again on the
  # final library -- round 6 changed the conv kernels' K order and XCD rectangles after the table was last tuned -- then same-box A/B
  python - <<PYEOF
import json, re
R, O = "$R", "$O"
d = json.load(open(f"{R}/diffusers_amd/tuned/gfx950.json"))
sdxl = set(json.load(open(f"{R}/tests/golden/sdxl_gemm_shape_keys.json"))["keys"])
drop = []
for k, v in d["entries"].items():
    if k not in sdxl or k.startswith(("pair:", "qkv:")) or d["tiles"][v[0]].startswith("k3:"):
        continue
    m = re.match(r"(conv\d|lin):M(\d+):", k)
    if m and int(m.group(2)) <= 32768 and ":N16384:" not in k and ":K16384:" not in k:
        drop.append(k)
for k in drop:
    del d["entries"][k]
json.dump(d, open(f"{O}/table_pruned_sdxl.json", "w"))
print("entries to tune again:", len(drop))
PYEOF
  cp $O/table_pruned_sdxl.json $O/table_retuned_sdxl.json
  DIFFUSERS_AMD_GEMM_FAMILY=all DIFFUSERS_AMD_TUNE_DB=$O/table_retuned_sdxl.json DIFFUSERS_AMD_TUNE_SAVE=$O/table_retuned_sdxl.json timeout 1200 python bench.py --steps 1 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/retune_sdxl.json 2> $O/retune_sdxl.err; echo "retune sdxl rc=$? $(grep -o '"value": [0-9.]*' $O/retune_sdxl.json | head -1) $(grep -o '"tuned_live": [0-9]*' $O/retune_sdxl.json)"
  python - <<PYEOF
import json
a = json.load(open("$R/diffusers_amd/tuned/gfx950.json"))["entries"]
b = json.load(open("$O/table_retuned_sdxl.json"))
T = b["tiles"]
n = 0
for k, v in sorted(b["entries"].items()):
    o = a.get(k)
    if o is None or o[:2] != v[:2] or o[3:] != v[3:]:
        n += 1
        print(f"  {k:58s} {T[o[0]] if o else '-':11s} st{o[1] if o else '-'} {o[2] if o else 0:7.1f}us -> {T[v[0]]:11s} st{v[1]} {v[2]:7.1f}us")
print("changed:", n)
PYEOF
  for rep in 1 2 3; do
    for tb in $R/diffusers_amd/tuned/gfx950.json $O/table_retuned_sdxl.json; do
      DIFFUSERS_AMD_TUNE_DB=$tb timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/ab_sdxl.json 2> $O/ab_sdxl.err; echo "$(basename $tb) sdxl rc=$? $(grep -o '"value": [0-9.]*' $O/ab_sdxl.json | head -1) $(grep -o '"tuned_live": [0-9]*' $O/ab_sdxl.json)"
    done
  done
fi
if [[ $WHAT == *tunestep* ]]; then
  # third session: the step itself as the tuner's arbiter (tools/insitu_tune.py), then bench.py with the shipped and the in-situ table
  timeout 1800 python tools/insitu_tune.py $O/r06j_insitu_tune.json ${INSITU_KEYS:-16} 16 0.15 ${INSITU_CAP:-1100} ${INSITU_SKIP:-} > $O/insitu_tune.log 2>&1; echo "insitu tune rc=$?"
  grep "^\[insitu\]" $O/insitu_tune.log | grep -v "/st[0-9]*: " | cut -c1-260 | tail -60
  if [[ -f $O/table_insitu.json ]]; then
    for rep in 1 2 3; do
      for tb in $R/diffusers_amd/tuned/gfx950.json $O/table_insitu.json; do
        DIFFUSERS_AMD_TUNE_DB=$tb timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/ab_sdxl.json 2> $O/ab_sdxl.err; echo "$(basename $tb) sdxl rc=$? $(grep -o '"value": [0-9.]*' $O/ab_sdxl.json | head -1) $(grep -o '"tuned_live": [0-9]*' $O/ab_sdxl.json)"
      done
    done
  fi
fi
if [[ $WHAT == *tuneother* ]]; then
  # the same arbiter for the SD1.5 / DDPM configurations (their own entries only; chained: the second pass starts from the first one's table)
  TB=$R/diffusers_amd/tuned/gfx950.json
  for cfg in sd15 ddpm; do
    DIFFUSERS_AMD_TUNE_DB=$TB timeout 1200 python tools/insitu_tune.py $O/r06j_insitu_tune_$cfg.json ${INSITU_KEYS:-20} 12 0.3 ${INSITU_CAP:-500} - $cfg > $O/insitu_tune_$cfg.log 2>&1; echo "insitu tune $cfg rc=$?"
    grep "^\[insitu\]" $O/insitu_tune_$cfg.log | grep -v "/st[0-9]*: \|\] key " | cut -c1-260 | tail -40
    [[ -f $O/table_insitu_$cfg.json ]] && TB=$O/table_insitu_$cfg.json
  done
  cp $TB $O/table_insitu_other.json
  for rep in 1 2; do
    for tb in $R/diffusers_amd/tuned/gfx950.json $O/table_insitu_other.json; do
      for cfg in sd15 ddpm; do
        DIFFUSERS_AMD_TUNE_DB=$tb timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/ab_$cfg.json 2> $O/ab_$cfg.err; echo "$(basename $tb) $cfg rc=$? $(grep -o '"value": [0-9.]*' $O/ab_$cfg.json | head -1) $(grep -o '"tuned_live": [0-9]*' $O/ab_$cfg.json) $(grep -o '"psnr[a-z_]*": [0-9.]*' $O/ab_$cfg.json | head -2 | tr '\n' ' ')"
      done
    done
  done
fi
if [[ $WHAT == *tuneflux* ]]; then
  DIFFUSERS_AMD_TUNE_DB=$R/diffusers_amd/tuned/gfx950.json timeout 1500 python tools/insitu_tune.py $O/r06j_insitu_tune_flux.json ${INSITU_KEYS:-16} 8 0.3 ${INSITU_CAP:-600} - flux > $O/insitu_tune_flux.log 2>&1; echo "insitu tune flux rc=$?"
  grep "^\[insitu\]" $O/insitu_tune_flux.log | grep -v "/st[0-9]*: " | cut -c1-260 | tail -50
  if [[ -f $O/table_insitu_flux.json ]]; then
    for rep in 1 2; do
      for tb in $R/diffusers_amd/tuned/gfx950.json $O/table_insitu_flux.json; do
        DIFFUSERS_AMD_TUNE_DB=$tb timeout 600 python bench.py --config flux --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/ab_flux.json 2> $O/ab_flux.err; echo "$(basename $tb) flux rc=$? $(grep -o '"value": [0-9.]*' $O/ab_flux.json | head -1) $(grep -o '"tuned_live": [0-9]*' $O/ab_flux.json)"
      done
    done
  fi
fi
if [[ $WHAT == *attnsplit* ]]; then
  timeout 900 python -m pytest tests/test_attention_split.py -q -s --timeout 600 > $O/pytest_attnsplit.log 2>&1; echo "pytest attnsplit rc=$?"
  grep -E "passed|failed|FAILED|Error|assert|\[split\]" $O/pytest_attnsplit.log | tail -40
  timeout 600 python tools/bench_attn_r6.py $O/r06_attention.jsonl > $O/attn_r6.log 2>&1; echo "attn bench rc=$?"; cat $O/attn_r6.log | cut -c1-300 | tail -40
fi
if [[ $WHAT == *attntests* ]]; then
  timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_attention_boundary.py -m gpu -q --timeout 600 -k "attention or attn or backend or processor" > $O/pytest_attn.log 2>&1; echo "pytest attn rc=$?"
  grep -E "passed|failed|FAILED|Error|assert" $O/pytest_attn.log | tail -20
fi
if [[ $WHAT == *newparity* ]]; then
  timeout 2400 python -m pytest tests/test_full_size_gpu.py -m gpu -q -s --timeout 1800 -k "ddpm_cat or 3_step" > $O/pytest_newparity.log 2>&1; echo "pytest newparity rc=$?"
  grep -E "passed|failed|FAILED|Error|assert|\[parity\]" $O/pytest_newparity.log | tail -20
fi
if [[ $WHAT == *splitab* ]]; then
  for m in 0 1 0 1; do
    DIFFUSERS_AMD_ATTN_SPLIT=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_split$m.json 2> $O/bench_split$m.err; echo "attn split $m rc=$? $(cut -c1-140 $O/bench_split$m.json | grep -o '"value": [0-9.]*')"
  done
fi
if [[ $WHAT == *benchfast* ]]; then
  timeout 900 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  cut -c1-1400 $O/bench.json; grep "^\[bench" $O/bench.err | tail -20
fi
if [[ $WHAT == *fulltest* ]]; then
  timeout 3600 python -m pytest tests -m gpu -q -s --timeout 1800 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -20
  grep -E "\[parity\] (SDXL|FLUX|Wan|SD1.5|full|Auto|ddpm)|\[drop-in\]|\[B3\]|\[B4\]|\[rccl\]|\[split\]" $O/pytest_gpu.log | tail -70
fi
if [[ $WHAT == *benchfull* && $WHAT != *trafficfirst* ]]; then
  timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
  cut -c1-300 $O/bench_full.json; grep "^\[bench" $O/bench_full.err | tail -40
fi
if [[ $WHAT == *traffic* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/pmc_traffic; mkdir -p $O/pmc_traffic
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_traffic/fetch -o sdxl -- python $R/tools/pmc_one_step.py 2 $O/pmc_traffic/launch_log.json > $O/pmc_traffic/fetch.log 2>&1; echo "pmc fetch rc=$?"
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_traffic/write -o sdxl -- python $R/tools/pmc_one_step.py 2 > $O/pmc_traffic/write.log 2>&1; echo "pmc write rc=$?"
  cd $R
  ALGO=$(python -c "import json;print(json.load(open('$O/bench_full.json'))['roofline']['algorithmic_bytes_per_launch'])" 2>/dev/null || python -c "import json;print(json.load(open('$R/profiles/r05r_bench_line_full.json'))['roofline']['algorithmic_bytes_per_launch'])" 2>/dev/null)
  python tools/pmc_traffic.py $O/pmc_traffic/fetch $O/pmc_traffic/write $O/r06_sdxl_traffic.md $O/sdxl_traffic.json "$ALGO" 140 $O/pmc_traffic/launch_log.json 2
  cp $O/sdxl_traffic.json $R/profiles/sdxl_traffic.json   # a later `benchfull` stage of this call reads it (roofline.traffic)
  find $O/pmc_traffic -name '*kernel_trace*' -delete
  find $O/pmc_traffic -name '*counter_collection.csv' -size +8M -delete
  tail -4 $O/pmc_traffic/fetch.log | cut -c1-200
fi
if [[ $WHAT == *prof* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof
  timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o sdxl -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference --no-other-configs > $O/prof.log 2>&1; echo "prof rc=$?"
  grep '"metric"' $O/prof.log | cut -c1-200
  find $O/prof -name '*kernel_trace*' -size +30M -delete
  cd $R
  python tools/prof_summary.py $(find $O/prof -name '*kernel_stats.csv' | head -1) "r06 sdxl bench (--steps 1 --warmup 1)" > $O/prof_summary.md 2>> $O/prof.log; head -60 $O/prof_summary.md
fi
if [[ $WHAT == *trafficfirst* && $WHAT == *benchfull* ]]; then
  timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
  cut -c1-300 $O/bench_full.json; grep "^\[bench" $O/bench_full.err | tail -40
fi
if [[ $WHAT == *otherbench* ]]; then
  for c in sd15 flux ddpm; do
    timeout 900 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$? $(cut -c1-200 $O/bench_$c.json)"
  done
fi
if [[ $WHAT == *insitu* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/pmc_insitu; mkdir -p $O/pmc_insitu
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum -f csv -d $O/pmc_insitu/a -o sdxl -- python $R/tools/pmc_one_step.py 2 $O/pmc_insitu/launch_log.json > $O/pmc_insitu/a.log 2>&1; echo "pmc insitu a rc=$?"
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES -f csv -d $O/pmc_insitu/b -o sdxl -- python $R/tools/pmc_one_step.py 2 > $O/pmc_insitu/b.log 2>&1; echo "pmc insitu b rc=$?"
  cd $R
  python tools/pmc_insitu.py $O/pmc_insitu/launch_log.json 140 $O/r06_insitu_counters.md $O/pmc_insitu/a $O/pmc_insitu/b
  find $O/pmc_insitu -name '*.csv' -size +8M -delete
  tail -3 $O/pmc_insitu/a.log | cut -c1-200
fi
if [[ $WHAT == *gnmulti* ]]; then
  timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s --timeout 600 -k "groupnorm" > $O/pytest_gn.log 2>&1; echo "pytest gn rc=$?"
  grep -E "passed|failed|FAILED|Error|assert" $O/pytest_gn.log | tail -20
  timeout 600 python tools/bench_norms_r6.py $O/r06_groupnorm_several_workgroups.jsonl > $O/norms_r6.log 2>&1; echo "norms bench rc=$?"; cut -c1-260 $O/norms_r6.log | tail -36
fi
if [[ $WHAT == *gnab* ]]; then
  for m in 0 1 0 1; do
    DIFFUSERS_AMD_GN_MULTI=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_gnm$m.json 2> $O/bench_gnm$m.err; echo "gn multi $m rc=$? $(cut -c1-140 $O/bench_gnm$m.json | grep -o '"value": [0-9.]*')"
  done
fi
if [[ $WHAT == *ring4ab* ]]; then
  for m in 0 1 0 1; do
    DA_ATTN2_RING4=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_ring$m.json 2> $O/bench_ring$m.err; echo "attn ring4 $m rc=$? $(cut -c1-140 $O/bench_ring$m.json | grep -o '"value": [0-9.]*')"
  done
fi
if [[ $WHAT == *inflight* ]]; then
  timeout 900 python tools/bench_inflight.py $O/r06_two_in_flight.json 3 > $O/inflight.log 2>&1; echo "inflight rc=$?"; tail -5 $O/inflight.log | cut -c1-300
fi
if [[ $WHAT == *convchunk* ]]; then
  timeout 1200 python -m pytest tests/test_gemm_k2_gpu.py tests/test_kernels_gpu.py tests/test_models_gpu.py -m gpu -q --timeout 900 -k "conv or unet or vae or k2 or resnet" > $O/pytest_conv.log 2>&1; echo "pytest conv rc=$?"
  grep -E "passed|failed|FAILED|Error|assert" $O/pytest_conv.log | tail -20
  for m in 0 128 256 0 128 512; do
    DA_CONV_CHUNK=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_cc$m.json 2> $O/bench_cc$m.err; echo "conv chunk $m rc=$? $(cut -c1-140 $O/bench_cc$m.json | grep -o '"value": [0-9.]*')"
  done
fi
if [[ $WHAT == *convin* ]]; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -k "conv_thin" > $O/pytest_convin.log 2>&1; echo "pytest conv_thin rc=$?"
  grep -E "passed|failed|FAILED|Error|assert" $O/pytest_convin.log | tail -8
  for m in 0 1 0 1; do
    DA_CONV_IN_QUAD=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_ci$m.json 2> $O/bench_ci$m.err; echo "conv_in quad $m rc=$? $(cut -c1-140 $O/bench_ci$m.json | grep -o '"value": [0-9.]*')"
  done
fi
if [[ $WHAT == *dbginflight* ]]; then
  timeout 600 python tools/debug_inflight.py > $O/dbg_inflight_a.log 2>&1; echo "dbg a rc=$?"; grep -E "sequential|concurrent|Error" $O/dbg_inflight_a.log | cut -c1-200
  timeout 600 python tools/debug_inflight.py --headline-first > $O/dbg_inflight_b.log 2>&1; echo "dbg b rc=$?"; grep -E "sequential|concurrent|Error" $O/dbg_inflight_b.log | cut -c1-200
  DIFFUSERS_AMD_ATTN_SPLIT=0 timeout 600 python tools/debug_inflight.py --headline-first > $O/dbg_inflight_c.log 2>&1; echo "dbg c (no attn split) rc=$?"; grep -E "sequential|concurrent|Error" $O/dbg_inflight_c.log | cut -c1-200
  timeout 600 python tools/debug_inflight.py --headline-first --latent > $O/dbg_inflight_d.log 2>&1; echo "dbg d (latents) rc=$?"; grep -E "sequential|concurrent|Error" $O/dbg_inflight_d.log | cut -c1-200
fi
if [[ $WHAT == *dbgdecode* ]]; then
  timeout 300 python tools/debug_decode_concurrent.py 2>&1 | grep -E "RESULT|Error" | cut -c1-300
  timeout 300 python tools/debug_decode_concurrent.py --nosplitk 2>&1 | grep -E "RESULT|Error|switched" | cut -c1-300
  DA_GN_FUSED=0 timeout 300 python tools/debug_decode_concurrent.py 2>&1 | grep -E "RESULT|Error" | cut -c1-300
  DIFFUSERS_AMD_PREFETCH=0 timeout 300 python tools/debug_decode_concurrent.py 2>&1 | grep -E "RESULT|Error" | cut -c1-300
  DA_CONV_IN_QUAD=0 timeout 300 python tools/debug_decode_concurrent.py --raw 2>&1 | grep -E "RESULT|Error" | cut -c1-300
fi
if [[ $WHAT == *dbgops* ]]; then
  timeout 600 python tools/debug_ops_concurrent.py 2>&1 | grep -E "RESULT|Error|splitk" | cut -c1-200
fi
if [[ $WHAT == *dbgalloc* ]]; then
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 600 python tools/debug_decode_concurrent.py 2>&1 | grep -E "RESULT|Error" | cut -c1-300
  PYTORCH_NO_HIP_MEMORY_CACHING=1 timeout 600 python tools/debug_decode_concurrent.py 2>&1 | grep -E "RESULT|Error" | cut -c1-300
  AMD_SERIALIZE_KERNEL=3 timeout 600 python tools/debug_decode_concurrent.py 2>&1 | grep -E "RESULT|Error" | cut -c1-300
  GPU_MAX_HW_QUEUES=1 timeout 600 python tools/debug_decode_concurrent.py 2>&1 | grep -E "RESULT|Error" | cut -c1-300
fi
if [[ $WHAT == *dbgtrace* ]]; then
  timeout 600 python tools/debug_decode_trace.py 2>&1 | grep -E "RESULT|Error" | cut -c1-260
fi
if [[ $WHAT == *dbgvs* ]]; then
  timeout 900 python tools/debug_decode_vs_op.py 2>&1 | grep -E "RESULT B|Error" | cut -c1-260
fi
if [[ $WHAT == *dbglock* ]]; then
  timeout 600 python tools/debug_inflight.py --headline-first > $O/dbg_inflight_lock.log 2>&1; echo "dbg lock rc=$?"; grep -E "sequential|concurrent|Error" $O/dbg_inflight_lock.log | cut -c1-200
  timeout 600 python tools/debug_inflight.py > $O/dbg_inflight_lock2.log 2>&1; echo "dbg lock2 rc=$?"; grep -E "sequential|concurrent|Error" $O/dbg_inflight_lock2.log | cut -c1-200
  timeout 900 python tools/bench_inflight.py $O/r06_two_in_flight.json 3 > $O/inflight.log 2>&1; echo "inflight rc=$?"; tail -3 $O/inflight.log | cut -c1-300
fi
if [[ $WHAT == *dbggraph* ]]; then
  timeout 600 python tools/debug_decode_graph_concurrent.py 2>&1 | grep -E "RESULT|round|sequential|Error" | cut -c1-200
fi
if [[ $WHAT == *dbgoob* ]]; then
  timeout 900 python tools/debug_oob.py 2>&1 | grep -E "RESULT|OOB|alloc |Error|decode alloc" | cut -c1-260 | head -60
  timeout 900 python tools/debug_oob.py --unet --gb 64 2>&1 | grep -E "RESULT|OOB|alloc |Error" | cut -c1-260 | head -60
fi
if [[ $WHAT == *dbglaunch* ]]; then
  timeout 300 python tools/debug_decode_concurrent.py --launchlock 2>&1 | grep -E "RESULT|Error|installed" | cut -c1-300
  HIP_FORCE_DEV_KERNARG=0 timeout 300 python tools/debug_decode_concurrent.py 2>&1 | grep -E "RESULT|Error" | cut -c1-300
  HIP_FORCE_DEV_KERNARG=1 timeout 300 python tools/debug_decode_concurrent.py 2>&1 | grep -E "RESULT|Error" | cut -c1-300
fi
if [[ $WHAT == *dbgarena* ]]; then
  timeout 300 python tools/debug_decode_concurrent.py --arena 2>&1 | grep -E "RESULT|Error|installed" | cut -c1-300
fi
if [[ $WHAT == *r05ab* ]]; then
  # the round-5 library (git edaa4bd, built in the worktree _r05_ab/) against this round's, alternating, one box
  for m in r05 r06 r05 r06; do
    if [[ $m == r05 ]]; then D=$R/_r05_ab; else D=$R; fi
    (cd $D && timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-other-configs > $O/bench_ab_$m.json 2> $O/bench_ab_$m.err); echo "library $m rc=$? $(cut -c1-140 $O/bench_ab_$m.json | grep -o '"value": [0-9.]*') $(python -c "import json;d=json.load(open('$O/bench_ab_$m.json'));r=d['roofline'];print('igemm',round(r['frac'],3),'attention',round([k for k in r['kernels'] if k['name']=='attention'][0]['frac'],3), 'attn_ms', round([k for k in r['kernels'] if k['name']=='attention'][0]['ms'],3))" 2>/dev/null)"
  done
  for c in sd15 flux; do
    for m in r05 r06; do
      if [[ $m == r05 ]]; then D=$R/_r05_ab; else D=$R; fi
      (cd $D && timeout 600 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_ab_${c}_$m.json 2> $O/bench_ab_${c}_$m.err); echo "$c library $m rc=$? $(cut -c1-160 $O/bench_ab_${c}_$m.json | grep -o '"value": [0-9.]*')"
    done
  done
fi
if [[ $WHAT == *anyorder* ]]; then
  hipcc --offload-arch=gfx950 -O2 -Iinclude tools/probe_anyorder.cpp -Ldiffusers_amd/_C -ldiffusers_amd -Wl,-rpath,$R/diffusers_amd/_C -o /tmp/probe_anyorder 2> $O/probe_anyorder_build.log; echo "probe build rc=$?"
  timeout 300 /tmp/probe_anyorder > $O/r06c_anyorder_probe.jsonl 2> $O/probe_anyorder.err; echo "probe rc=$?"; cat $O/r06c_anyorder_probe.jsonl | cut -c1-220
fi
if [[ $WHAT == *convknobs* ]]; then
  cd /tmp && export TMPDIR=/tmp
  for v in base chunkauto xcdconv both; do
    case $v in base) E="";; chunkauto) E="DA_CONV_CHUNK=auto";; xcdconv) E="DA_XCD_CONV=1";; both) E="DA_CONV_CHUNK=auto DA_XCD_CONV=1";; esac
    rm -rf $O/pmc_ck_$v; mkdir -p $O/pmc_ck_$v
    env $E DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -f csv -d $O/pmc_ck_$v/a -o sdxl -- python $R/tools/pmc_one_step.py 2 $O/pmc_ck_$v/launch_log.json > $O/pmc_ck_$v/a.log 2>&1; echo "pmc $v rc=$?"
    (cd $R && python tools/pmc_insitu.py $O/pmc_ck_$v/launch_log.json 140 $O/r06c_insitu_$v.md $O/pmc_ck_$v/a > /dev/null 2>> $O/pmc_ck_$v/a.log)
    find $O/pmc_ck_$v -name '*.csv' -size +4M -delete
  done
  cd $R
  python - <<'PY'
import json, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out"
tabs = {v: json.load(open(f"{O}/r06c_insitu_{v}.json")) for v in ("base", "chunkauto", "xcdconv", "both") if os.path.exists(f"{O}/r06c_insitu_{v}.json")}
if "base" in tabs:
    print("| population | " + " | ".join(f"{v} us / MB fetched" for v in tabs) + " |")
    for k in tabs["base"]:
        if "conv" not in k.split("::")[0]:
            continue
        cells = []
        for v, t in tabs.items():
            m = [e for kk, e in t.items() if kk.split("::")[0] == k.split("::")[0]]
            cells.append(f"{m[0]['us_profiled']:.1f} / {m[0].get('TCC_EA0_RDREQ_sum', 0) * 128 / 1e6:.0f}" if m else "-")
        print(f"| {k.split('::')[0]} | " + " | ".join(cells) + " |")
PY
fi
if [[ $WHAT == *convab* ]]; then
  for pass in 1 2; do
  for v in base chunkauto xcdconv both; do
    case $v in base) E="";; chunkauto) E="DA_CONV_CHUNK=auto";; xcdconv) E="DA_XCD_CONV=1";; both) E="DA_CONV_CHUNK=auto DA_XCD_CONV=1";; esac
    env $E timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_ck_$v.json 2> $O/bench_ck_$v.err; echo "conv knobs $v rc=$? $(cut -c1-140 $O/bench_ck_$v.json | grep -o '"value": [0-9.]*')"
  done
  done
fi
if [[ $WHAT == *lnf3test* ]]; then
  timeout 1500 python -m pytest tests/test_gemm_k3_gpu.py tests/test_gemm_k2_gpu.py -m gpu -q --timeout 900 -k "geglu or fold or LN or ln" > $O/pytest_lnf3.log 2>&1; echo "pytest lnf3 rc=$?"
  grep -E "passed|failed|FAILED|Error|assert" $O/pytest_lnf3.log | tail -20
fi
if [[ $WHAT == *lnf3ab* ]]; then
  for pass in 1 2; do
  for m in 2 1; do
    DIFFUSERS_AMD_LN_FOLD=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_lnf$m.json 2> $O/bench_lnf$m.err; echo "LN_FOLD=$m rc=$? $(cut -c1-140 $O/bench_lnf$m.json | grep -o '"value": [0-9.]*')"
  done
  done
  DIFFUSERS_AMD_LN_FOLD=1 DIFFUSERS_AMD_LN_FOLD_K3=0 timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_lnf1k1.json 2> $O/bench_lnf1k1.err; echo "LN_FOLD=1 on k1:128x320 rc=$? $(cut -c1-140 $O/bench_lnf1k1.json | grep -o '"value": [0-9.]*')"
  DIFFUSERS_AMD_LN_FOLD=1 timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-live-traffic > $O/bench_lnf1_parity.json 2> $O/bench_lnf1_parity.err; echo "LN_FOLD=1 with parity rc=$?"; grep -E "parity:|dropin:|timed region done" $O/bench_lnf1_parity.err | cut -c1-200
  for c in sd15; do
    for m in 2 1; do
      DIFFUSERS_AMD_LN_FOLD=$m timeout 900 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_${c}_lnf$m.json 2> $O/bench_${c}_lnf$m.err; echo "$c LN_FOLD=$m rc=$? $(cut -c1-200 $O/bench_${c}_lnf$m.json | grep -o '"value": [0-9.]*')"
    done
  done
fi
if [[ $WHAT == *driverline* ]]; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; echo "smoke rc=$? $(tail -1 $O/smoke.log | cut -c1-120)"
  T0=$(date +%s); timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; echo "bench (driver protocol) rc=$? in $(( $(date +%s) - T0 )) s"
  cut -c1-260 $O/bench_driver.json; grep "^\[bench" $O/bench_driver.err | grep -E "timed region done|traffic|parity:|other_configs:" | cut -c1-200
fi
if [[ $WHAT == *ldscanary* ]]; then
  timeout 900 python tools/debug_lds_canary.py $O/r06f_lds_canary.jsonl > $O/lds_canary.log 2>&1; echo "lds canary rc=$?"; grep -E "RESULT|Error|error" $O/lds_canary.log | cut -c1-330
fi
if [[ $WHAT == *rowtail* ]]; then
  timeout 600 python tools/bench_rowtail.py $O/r06f_row_tail_split.jsonl > $O/rowtail.log 2>&1; echo "rowtail rc=$?"; tail -3 $O/rowtail.log | cut -c1-900
fi
if [[ $WHAT == *splitall* ]]; then
  for m in 0 2 3 4; do
    DA_ATTN_SPLIT_ALL=$m timeout 300 python tools/bench_attn_splitall.py 2>&1 | grep -E "^\{|Error" | cut -c1-420 | tee -a $O/r06g_attention_split_all.jsonl
  done
  DA_ATTN_SPLIT_ALL=2 timeout 900 python -m pytest tests/test_attention_split.py tests/test_kernels_gpu.py -m gpu -q --timeout 600 -k "attention or attn" > $O/pytest_splitall.log 2>&1; echo "pytest (split all 2) rc=$?"; grep -E "passed|failed|FAILED|Error|assert" $O/pytest_splitall.log | tail -8
  for pass in 1 2; do
  for m in 0 2 4; do
    DA_ATTN_SPLIT_ALL=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_sa$m.json 2> $O/bench_sa$m.err; echo "split all $m rc=$? $(cut -c1-140 $O/bench_sa$m.json | grep -o '"value": [0-9.]*')"
  done
  done
fi
if [[ $WHAT == *profsd15* ]]; then
  cd /tmp && export TMPDIR=/tmp
  for c in sd15 ddpm; do
    rm -rf $O/prof_$c
    timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $O/prof_$c -o $c -- python $R/bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-reference > $O/prof_$c.log 2>&1; echo "prof $c rc=$?"
    grep '"metric"' $O/prof_$c.log | cut -c1-200
    find $O/prof_$c -name '*kernel_trace*' -size +30M -delete
    (cd $R && python tools/prof_summary.py $(find $O/prof_$c -name '*kernel_stats.csv' | head -1) "r06 $c bench (--steps 1 --warmup 1)" > $O/prof_summary_$c.md 2>> $O/prof_$c.log); head -45 $O/prof_summary_$c.md | cut -c1-170
  done
  cd $R
fi
