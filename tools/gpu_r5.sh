#!/bin/bash
# Round-5 gpurun stages.  usage: gpu_r5.sh "boundary benchfast ..." -- stages run in FILE order, selected by substring (keep stage names
# free of each other as substrings).  Stages of the eight-phase GEMM work: k3test (tests/test_gemm_k3_gpu.py), k3bench (tools/bench_k3.py,
# K3_ONLY=<name filter>), k3retune (tools/retune_k3.py -> gpurun_out/table_k3.json), k3prio (priority forms in chains), prioinsitu
# (the same in situ), gegluab (new / old table on one box), forcedk3ab, attnab (flash attention priority pair), kvpfab (K / V^T prefetch),
# gemmtests, otherbench, wanfull; the round's record: fulltest, then "traffic prof trafficfirst" (PMC traffic, rocprofv3 stats, full
# default bench line reading the fresh traffic file).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
WHAT=${1:-boundary}
if [[ $WHAT == *boundary* ]]; then
  timeout 900 python -m pytest tests/test_attention_boundary.py tests/test_distributed_gpu.py -m gpu -q -s --timeout 600 > $O/pytest_boundary.log 2>&1; echo "pytest boundary rc=$?"
  grep -E "passed|failed|FAILED|Error|\[B3\]|\[B4\]|\[rccl\]" $O/pytest_boundary.log | tail -30
fi
if [[ $WHAT == *dropin* ]]; then
  timeout 900 python -m pytest tests/test_reference_dropin_gpu.py -m gpu -q -s --timeout 600 -k "flux or wan or sd_pipeline" > $O/pytest_dropin.log 2>&1; echo "pytest dropin rc=$?"
  grep -E "passed|failed|FAILED|Error|\[drop-in\]" $O/pytest_dropin.log | tail -30
  timeout 900 python -m pytest tests/test_full_size_gpu.py -m gpu -q -s --timeout 600 -k "wan_vae" > $O/pytest_wanvae.log 2>&1; echo "pytest wan vae rc=$?"
  grep -E "passed|failed|FAILED|Error|\[parity\]" $O/pytest_wanvae.log | tail -30
fi
if [[ $WHAT == *benchfast* ]]; then
  timeout 900 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-other-configs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  cut -c1-1400 $O/bench.json; grep "^\[bench" $O/bench.err | tail -20
fi
if [[ $WHAT == *wanbench* ]]; then
  timeout 900 python bench.py --config wan --steps 1 --warmup 1 --denoise-steps ${WAN_STEPS:-6} --no-cpu-baseline > $O/bench_wan.json 2> $O/bench_wan.err; echo "wan rc=$?"
  cut -c1-1600 $O/bench_wan.json; tail -3 $O/bench_wan.err | cut -c1-300
fi
if [[ $WHAT == *fulltest* ]]; then
  timeout 2400 python -m pytest tests -m gpu -q -s --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -20
  grep -E "\[parity\] (SDXL|FLUX|Wan|SD1.5|full|Auto)|\[drop-in\]|\[B3\]|\[B4\]|\[rccl\]" $O/pytest_gpu.log | tail -60
fi
if [[ $WHAT == *benchfull* && $WHAT != *trafficfirst* ]]; then
  timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
  cut -c1-300 $O/bench_full.json; grep "^\[bench" $O/bench_full.err | tail -40
fi
if [[ $WHAT == *traffic* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/pmc_traffic; mkdir -p $O/pmc_traffic
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc_traffic/fetch -o sdxl -- python $R/tools/pmc_one_step.py 2 > $O/pmc_traffic/fetch.log 2>&1; echo "pmc fetch rc=$?"
  DIFFUSERS_AMD_TUNE=0 timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc_traffic/write -o sdxl -- python $R/tools/pmc_one_step.py 2 > $O/pmc_traffic/write.log 2>&1; echo "pmc write rc=$?"
  cd $R
  ALGO=$(python -c "import json;print(json.load(open('$O/bench_full.json'))['roofline']['algorithmic_bytes_per_launch'])" 2>/dev/null || python -c "import json;print(json.load(open('$R/profiles/r04r_bench_line_full.json'))['roofline']['algorithmic_bytes_per_launch'])" 2>/dev/null)
  python tools/pmc_traffic.py $O/pmc_traffic/fetch $O/pmc_traffic/write $O/r05_sdxl_traffic.md $O/sdxl_traffic.json "$ALGO" 140
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
  python tools/prof_summary.py $(find $O/prof -name '*kernel_stats.csv' | head -1) "r05 sdxl bench (--steps 1 --warmup 1)" > $O/prof_summary.md 2>> $O/prof.log; head -60 $O/prof_summary.md
fi
if [[ $WHAT == *xattn* ]]; then
  timeout 600 python -m pytest tests/test_xattn_gpu.py tests/test_attention_boundary.py -m gpu -q -s --timeout 300 -x > $O/pytest_xattn.log 2>&1; echo "pytest xattn rc=$?"
  grep -E "passed|failed|FAILED|Error|assert|\[parity\]|\[B3\]" $O/pytest_xattn.log | tail -40
  timeout 300 python tools/bench_xattn_r5.py $O/xattn_r5.jsonl > $O/xattn_r5.log 2>&1; echo "xattn bench rc=$?"; cat $O/xattn_r5.jsonl | cut -c1-400; tail -3 $O/xattn_r5.log | cut -c1-300
fi
if [[ $WHAT == *xab* ]]; then
  for m in 1 0 1 0; do
    DIFFUSERS_AMD_XATTN=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_xa$m.json 2> $O/bench_xa$m.err; echo "xattn $m rc=$? $(cut -c1-140 $O/bench_xa$m.json | grep -o '"value": [0-9.]*')"
  done
fi
if [[ $WHAT == *gnfused* ]]; then
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -s --timeout 300 -k "groupnorm" > $O/pytest_gn.log 2>&1; echo "pytest gn rc=$?"
  grep -E "passed|failed|FAILED|Error|assert" $O/pytest_gn.log | tail -20
  timeout 300 python tools/bench_norms_r5.py $O/norms_r5.jsonl > $O/norms_r5.log 2>&1; echo "norms bench rc=$?"; cat $O/norms_r5.jsonl | cut -c1-300
fi
if [[ $WHAT == *gnab* ]]; then
  for m in 1 0 1 0; do
    DA_GN_FUSED=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_gn$m.json 2> $O/bench_gn$m.err; echo "gn fused $m rc=$? $(cut -c1-140 $O/bench_gn$m.json | grep -o '"value": [0-9.]*')"
    DA_GN_FUSED=$m timeout 600 python bench.py --config sd15 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_sd15_gn$m.json 2> $O/bench_sd15_gn$m.err; echo "sd15 gn fused $m rc=$? $(cut -c1-160 $O/bench_sd15_gn$m.json | grep -o '"value": [0-9.]*')"
  done
fi
if [[ $WHAT == *splitk* ]]; then
  timeout 900 python tools/bench_splitk_r5.py $O/splitk_r5.jsonl > $O/splitk_r5.log 2>&1; echo "splitk bench rc=$?"; cat $O/splitk_r5.jsonl | cut -c1-700; tail -3 $O/splitk_r5.log | cut -c1-300
fi
if [[ $WHAT == *tuneconv* ]]; then
  DIFFUSERS_AMD_SPLITK=1 DIFFUSERS_AMD_GEMM_FAMILY=all python -c "import json;t=json.load(open('$R/diffusers_amd/tuned/gfx950.json'));t['entries']={k:v for k,v in t['entries'].items() if not k.startswith('conv3:M2048:N1280:C')};json.dump(t,open('$O/table_without_32x32_convs.json','w'))"; DIFFUSERS_AMD_TUNE_DB=$O/table_without_32x32_convs.json timeout 600 python tools/tune_shapes_r5.py > $O/tune_shapes_r5.log 2>&1; echo "tune rc=$?"; cat $O/tune_shapes_r5.log | cut -c1-300 | tail -12
fi
if [[ $WHAT == *attnprio* ]]; then
  rm -f $O/attn_r5.jsonl
  for pr in ${PRIOS:-0 1 2 3 4 5 0}; do DA_ATTN2_PRIO=$pr timeout 200 python tools/bench_attn_r5.py $O/attn_r5.jsonl > $O/attn_r5.log 2>&1; done
  cat $O/attn_r5.jsonl | cut -c1-200
fi
if [[ $WHAT == *prioab* ]]; then
  for m in 1 0 1 0; do
    DA_ATTN2_PRIO=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_prio$m.json 2> $O/bench_prio$m.err; echo "attn prio $m rc=$? $(cut -c1-140 $O/bench_prio$m.json | grep -o '"value": [0-9.]*')"
  done
  timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q --timeout 300 -k "attention" 2>&1 | tail -2
fi
if [[ $WHAT == *k3test* ]]; then
  timeout 600 python -m pytest tests/test_gemm_k3_gpu.py -m gpu -q --timeout 300 > $O/pytest_k3.log 2>&1; echo "pytest k3 rc=$?"
  grep -E "passed|failed|FAILED|Error|assert|differ" $O/pytest_k3.log | tail -20
fi
if [[ $WHAT == *k3bench* ]]; then
  rm -f $O/k3_r5.jsonl
  timeout 600 python tools/bench_k3.py $O/k3_r5.jsonl "${K3_ONLY:-}" > $O/k3_r5.log 2>&1; echo "k3 bench rc=$?"
  cat $O/k3_r5.jsonl | cut -c1-600; tail -3 $O/k3_r5.log | cut -c1-300
fi
if [[ $WHAT == *attnab* ]]; then
  rm -f $O/attn_r5.jsonl
  for pr in 0 1 0 1; do DA_ATTN2_PRIO=$pr timeout 200 python tools/bench_attn_r5.py $O/attn_r5.jsonl > $O/attn_r5.log 2>&1; done
  cat $O/attn_r5.jsonl | cut -c1-200
fi
if [[ $WHAT == *k3retune* ]]; then
  timeout 900 python tools/retune_k3.py $O/k3_retune.jsonl $O/table_k3.json > $O/k3_retune.log 2>&1; echo "k3 retune rc=$?"
  cut -c1-330 $O/k3_retune.log | tail -40
fi
if [[ $WHAT == *k3tabab* ]]; then
  # other configs before / after the table moved entries to k3 (same box): flux 4 steps, wan 6 steps (UniPC + decode), sdxl fast
  for tb in old new old new; do
    if [[ $tb == new ]]; then export DIFFUSERS_AMD_TUNE_DB=$O/table_k3.json; else unset DIFFUSERS_AMD_TUNE_DB; fi
    timeout 600 python bench.py --config flux --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_flux_$tb.json 2> $O/bench_flux_$tb.err; echo "flux $tb rc=$? $(grep -o '"value": [0-9.]*' $O/bench_flux_$tb.json | head -1)"
  done
  for tb in old new; do
    if [[ $tb == new ]]; then export DIFFUSERS_AMD_TUNE_DB=$O/table_k3.json; else unset DIFFUSERS_AMD_TUNE_DB; fi
    timeout 600 python bench.py --config wan --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline --no-roofline > $O/bench_wan_$tb.json 2> $O/bench_wan_$tb.err; echo "wan $tb rc=$? $(grep -o '"value": [0-9.]*' $O/bench_wan_$tb.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_wan_$tb.json | head -1)"
  done
  unset DIFFUSERS_AMD_TUNE_DB
fi
if [[ $WHAT == *kvpfab* ]]; then
  for m in 1 0 1 0; do
    DIFFUSERS_AMD_KV_PREFETCH=$m timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_kvpf$m.json 2> $O/bench_kvpf$m.err; echo "kv prefetch $m rc=$? $(cut -c1-140 $O/bench_kvpf$m.json | grep -o '"value": [0-9.]*')"
  done
fi
if [[ $WHAT == *gemmtests* ]]; then
  timeout 900 python -m pytest tests/test_gemm_k2_gpu.py tests/test_gemm_k3_gpu.py -m gpu -q --timeout 600 > $O/pytest_gemm.log 2>&1; echo "pytest gemm rc=$?"
  grep -E "passed|failed|FAILED|Error|assert|differ" $O/pytest_gemm.log | tail -20
fi
if [[ $WHAT == *otherbench* ]]; then
  timeout 600 python bench.py --config flux --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_flux.json 2> $O/bench_flux.err; echo "flux rc=$? $(grep -o '"value": [0-9.]*' $O/bench_flux.json | head -1)"
  timeout 600 python bench.py --config wan --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline > $O/bench_wan6.json 2> $O/bench_wan6.err; echo "wan rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_wan6.json | head -1)"
fi
if [[ $WHAT == *k3prio* ]]; then
  rm -f $O/k3_prio.jsonl
  for pr in 1 0 2 1 0 2; do echo "{\"DA_K3_PRIO\": $pr}" >> $O/k3_prio.jsonl; DA_K3_PRIO=$pr timeout 300 python tools/bench_k3.py $O/k3_prio.jsonl "${K3_ONLY:-square}" > $O/k3_prio.log 2>&1; done
  python - <<PY
import json
pr=None
for l in open("$O/k3_prio.jsonl"):
    d=json.loads(l)
    if "DA_K3_PRIO" in d: pr=d["DA_K3_PRIO"]; continue
    print(pr, d["name"], d.get("k3:256x256"), d.get("k1:256x256"))
PY
fi
if [[ $WHAT == *gegluab* ]]; then
  for tb in new old new old; do
    if [[ $tb == old ]]; then export DIFFUSERS_AMD_TUNE_DB=$R/profiles/r05n_table_before_geglu_tile.json; else unset DIFFUSERS_AMD_TUNE_DB; fi
    timeout 600 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/bench_geglu_$tb.json 2> $O/bench_geglu_$tb.err; echo "sdxl table $tb rc=$? $(cut -c1-140 $O/bench_geglu_$tb.json | grep -o '"value": [0-9.]*')"
    timeout 600 python bench.py --config sd15 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_sd15_$tb.json 2> $O/bench_sd15_$tb.err; echo "sd15 table $tb rc=$? $(cut -c1-160 $O/bench_sd15_$tb.json | grep -o '"value": [0-9.]*')"
  done
  unset DIFFUSERS_AMD_TUNE_DB
fi
if [[ $WHAT == *trafficfirst* ]]; then   # (the stages above run in file order: this one runs the full bench AFTER the traffic stage)
  timeout 1500 python bench.py --steps 3 --warmup 1 > $O/bench_full.json 2> $O/bench_full.err; echo "bench full rc=$?"
  cut -c1-300 $O/bench_full.json; grep "^\[bench" $O/bench_full.err | tail -40
fi
if [[ $WHAT == *wanfull* ]]; then
  timeout 900 python bench.py --config wan --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_wan_full.json 2> $O/bench_wan_full.err; echo "wan full rc=$? $(grep -o '"value": [0-9.]*' $O/bench_wan_full.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_wan_full.json | head -1)"
fi
if [[ $WHAT == *forcedk3ab* ]]; then
  for tb in new old new old; do
    if [[ $tb == new ]]; then export DIFFUSERS_AMD_TUNE_DB=$R/profiles/r05q_table_forced_k3_candidates.json; else unset DIFFUSERS_AMD_TUNE_DB; fi
    timeout 600 python bench.py --config flux --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/bench_flux_f$tb.json 2> $O/bench_flux_f$tb.err; echo "flux forced-k3 $tb rc=$? $(grep -o '"value": [0-9.]*' $O/bench_flux_f$tb.json | head -1)"
  done
  for tb in new old; do
    if [[ $tb == new ]]; then export DIFFUSERS_AMD_TUNE_DB=$R/profiles/r05q_table_forced_k3_candidates.json; else unset DIFFUSERS_AMD_TUNE_DB; fi
    timeout 600 python bench.py --config wan --steps 1 --warmup 1 --denoise-steps 6 --no-cpu-baseline --no-roofline > $O/bench_wan_f$tb.json 2> $O/bench_wan_f$tb.err; echo "wan forced-k3 $tb rc=$? $(grep -o '"ms_per_step": [0-9.]*' $O/bench_wan_f$tb.json | head -1)"
  done
  unset DIFFUSERS_AMD_TUNE_DB
fi
if [[ $WHAT == *prioinsitu* ]]; then
  for pr in 2 0 1 2 0 1; do
    DA_K3_PRIO=$pr timeout 300 python bench.py --steps 3 --warmup 1 --no-reference --no-cpu-baseline --no-roofline --no-other-configs > $O/b_sdxl_p$pr.json 2> $O/b_sdxl_p$pr.err; echo "sdxl DA_K3_PRIO=$pr $(cut -c1-140 $O/b_sdxl_p$pr.json | grep -o '"value": [0-9.]*')"
    DA_K3_PRIO=$pr timeout 300 python bench.py --config flux --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/b_flux_p$pr.json 2> $O/b_flux_p$pr.err; echo "flux DA_K3_PRIO=$pr $(grep -o '"value": [0-9.]*' $O/b_flux_p$pr.json | head -1)"
  done
fi
