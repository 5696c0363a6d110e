cd $GRAFT_REPO_ROOT
O=gpurun_out
T=tools/_trace_gemm2
{
$T 2048 1280 1280 10 2 1
$T 2048 1280 64 10 2 1
$T 2048 1280 5120 10 2 1
$T 2048 1280 1280 10 7 1
$T 8192 640 640 10 2 1
$T 2048 10240 1280 16 2 0
$T 2048 10240 1280 9 6 0
$T 2048 2560 1280 11 6 0
} > $O/trace_gemm2.md 2>&1
cat $O/trace_gemm2.md | head -150
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -m gpu -q -x --timeout 600 -k "postprocess or thin_out or sdxl_pipeline or tiny_sdxl or flux_pipeline" 2>&1 | tail -5
bash tools/gpu_r3.sh "benchfast traffic"
