#!/bin/bash
# One gpurun call: GPU parity tests, smoke, kernel microbenchmarks, bench line, rocprofv3 kernel stats.
# Everything lands in gpurun_out/.   usage: gpu_round.sh "test smoke kernels bench prof"
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
WHAT=${1:-test smoke kernels bench prof}
if [[ $WHAT == *retune* ]]; then
  # re-measure every GEMM variant choice with the current kernels: all workloads extend ONE table
  export DIFFUSERS_AMD_TUNE_DB=$O/tuned_all.json DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_all.json
  rm -f $O/tuned_all.json
fi
if [[ $WHAT == *test* ]]; then
  timeout 900 python -m pytest tests -m gpu -q -s --timeout 300 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  grep -E "^\[tune\]|passed|failed|FAILED|Error|\[parity\]" $O/pytest_gpu.log | tail -40
fi
if [[ $WHAT == *smoke* ]]; then
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
  tail -3 $O/smoke.log
fi
if [[ $WHAT == *kernels* ]]; then
  timeout 600 python tools/bench_kernels.py > $O/kernels.log 2>&1; echo "kernels rc=$?"
  grep -E '"op": "(linear|conv3x3).best"|"op": "(attention|groupnorm_silu|layernorm)"' $O/kernels.log | cut -c1-220
fi
if [[ $WHAT == *flux* ]]; then
  timeout 600 python tools/bench_flux.py > $O/flux.log 2>&1; echo "flux rc=$?"
  tail -4 $O/flux.log | cut -c1-300
fi
if [[ $WHAT == *wan* ]]; then
  timeout 900 python tools/bench_wan.py > $O/wan.log 2>&1; echo "wan rc=$?"
  tail -3 $O/wan.log | cut -c1-400
fi
if [[ $WHAT == *lastcall* ]]; then
  # end-of-round confirmation on the final build: whole GPU suite, smoke, then the secondary configs with the new kernels
  timeout 150 python -m pytest tests -m gpu -q -s --timeout 120 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  grep -E "passed|failed|FAILED|Error" $O/pytest_gpu.log | tail -5
  timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/smoke.log
  timeout 60 python tools/bench_wan_vae.py > $O/wan_vae.log 2>&1; echo "wan_vae rc=$?"; tail -1 $O/wan_vae.log | cut -c1-300
  timeout 60 python tools/bench_flux.py > $O/flux.log 2>&1; echo "flux rc=$?"; tail -2 $O/flux.log | cut -c1-300
  timeout 60 python tools/bench_sd15.py > $O/sd15.log 2>&1; echo "sd15 rc=$?"; tail -1 $O/sd15.log | cut -c1-300
fi
if [[ $WHAT == *final* ]]; then
  # re-measure the variant choice of every SDXL shape with the current kernels, merge over the shipped table, then the
  # bench line and the rocprofv3 kernel statistics of that build
  timeout 120 python -m pytest tests -m gpu -q --timeout 100 -k "gate_and_row_bias or accumulate_in_place" > $O/pytest_final.log 2>&1; echo "pytest final rc=$?"; tail -2 $O/pytest_final.log
  rm -f $O/tuned_sdxl.json
  DIFFUSERS_AMD_TUNE_DB=$O/none.json DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_sdxl.json timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/retune.json 2> $O/retune.err; echo "retune rc=$?"
  python - <<PYEOF
import json
a = json.load(open("$R/diffusers_amd/tuned/gfx950.json"))
b = json.load(open("$O/tuned_sdxl.json"))
ch = sum(1 for k, v in b["entries"].items() if k in a["entries"] and a["entries"][k][:2] != v[:2])
a["entries"].update(b["entries"])
json.dump(a, open("$O/tuned_merged.json", "w"), indent=0)
print("retuned", len(b["entries"]), "shapes;", ch, "changed variant; table now", len(a["entries"]))
PYEOF
  DIFFUSERS_AMD_TUNE_DB=$O/tuned_merged.json timeout 400 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  cat $O/bench.json
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof
  DIFFUSERS_AMD_TUNE_DB=$O/tuned_merged.json timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o sdxl -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1; echo "prof rc=$?"
  grep '"metric"' $O/prof.log | cut -c1-200
  find $O/prof -name '*kernel_trace*' -size +30M -delete
  cd $R
fi
if [[ $WHAT == *abstage* ]]; then
  # A/B of the GEMM staging modes on the whole SDXL image: buffer-addressed LDS-DMA (default) vs per-lane pointers
  timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/ab_buffer.json 2> $O/ab_buffer.err; echo "ab buffer rc=$?"
  DA_GEMM_FLAT_STAGING=1 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $O/ab_flat.json 2> $O/ab_flat.err; echo "ab flat rc=$?"
  cut -c1-160 $O/ab_buffer.json; cut -c1-160 $O/ab_flat.json
fi
if [[ $WHAT == *wanvae* ]]; then
  timeout 300 python -m pytest tests -m gpu -q -s --timeout 200 -k "rmsnorm_channels or permute_0213 or accumulate_in_place or wan_vae" > $O/pytest_wanvae.log 2>&1; echo "pytest wanvae rc=$?"
  grep -E "passed|failed|FAILED|Error|\[parity\]" $O/pytest_wanvae.log | tail -20
  DIFFUSERS_AMD_TUNE_SAVE=$O/tuned_wanvae.json timeout 400 python tools/bench_wan_vae.py > $O/wan_vae.log 2>&1; echo "wan_vae rc=$?"
  tail -5 $O/wan_vae.log | cut -c1-400
fi
if [[ $WHAT == *ddpm* ]]; then
  timeout 600 python tools/bench_ddpm.py > $O/ddpm.log 2>&1; echo "ddpm rc=$?"
  tail -3 $O/ddpm.log | cut -c1-400
fi
if [[ $WHAT == *sd15* ]]; then
  timeout 600 python tools/bench_sd15.py > $O/sd15.log 2>&1; echo "sd15 rc=$?"
  tail -3 $O/sd15.log | cut -c1-300
fi
if [[ $WHAT == *bench* ]]; then
  timeout 600 python bench.py --steps 3 --warmup 1 --save-tuning $O/tuned_gfx950.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
  cat $O/bench.json; grep "^\[bench" $O/bench.err | tail -12
fi
if [[ $WHAT == *prof* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/prof
  DIFFUSERS_AMD_TUNE_DB=$O/tuned_gfx950.json timeout 420 rocprofv3 --kernel-trace --stats -f csv -d $O/prof -o sdxl -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > $O/prof.log 2>&1; echo "prof rc=$?"
  grep '"metric"' $O/prof.log | cut -c1-200
  find $O/prof -type f | head
  find $O/prof -name '*kernel_trace*' -size +30M -delete
  cd $R
fi
if [[ $WHAT == *pmc* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 -L > $O/counters_list.txt 2>&1
  rm -rf $O/pmc; mkdir -p $O/pmc
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -f csv -d $O/pmc/sq -o focus -- python $R/tools/bench_focus.py > $O/pmc/sq.log 2>&1; echo "pmc sq rc=$?"
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum -f csv -d $O/pmc/fetch -o focus -- python $R/tools/bench_focus.py > $O/pmc/fetch.log 2>&1; echo "pmc fetch rc=$?"
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_MISS_sum TCC_REQ_sum -f csv -d $O/pmc/write -o focus -- python $R/tools/bench_focus.py > $O/pmc/write.log 2>&1; echo "pmc write rc=$?"
  export DIFFUSERS_AMD_TUNE_DB=$O/tuned_gfx950.json
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $O/pmc/bench_fetch -o sdxl -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $O/pmc/bench_fetch.log 2>&1; echo "pmc bench fetch rc=$?"
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $O/pmc/bench_write -o sdxl -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline > $O/pmc/bench_write.log 2>&1; echo "pmc bench write rc=$?"
  for d in sq fetch write bench_fetch bench_write; do python $R/tools/pmc_agg.py $O/pmc/$d > $O/pmc/$d.summary.csv 2>> $O/pmc/agg.err; done
  find $O/pmc -name '*kernel_trace*' -delete
  find $O/pmc -name '*counter_collection.csv' -size +8M -delete
  ls -la $O/pmc/*
  cd $R
fi
if [[ $WHAT == *lds2* ]]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $O/pmc2; mkdir -p $O/pmc2
  run_pass() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" -f csv -d $O/pmc2/$name -o focus -- python $R/tools/bench_focus.py > $O/pmc2/$name.log 2>&1; echo "pmc2 $name rc=$?"; python $R/tools/pmc_agg.py $O/pmc2/$name > $O/pmc2/$name.summary.csv 2>> $O/pmc2/agg.err; }
  run_pass lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES
  run_pass tcplat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
  run_pass tcpstall TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum
  run_pass ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
  find $O/pmc2 -name '*kernel_trace*' -delete
  find $O/pmc2 -name '*counter_collection.csv' -delete
  tail -2 $O/pmc2/*.log | cut -c1-200
  cd $R
fi
