set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_attention_fp8_gpu.py tests/test_attention_gpu.py tests/test_driver_gpu.py -x -q -m gpu > gpurun_out/r3c_tests5.log 2>&1 || { tail -40 gpurun_out/r3c_tests5.log; exit 1; }
tail -2 gpurun_out/r3c_tests5.log
timeout -k 10 300 python tools/bench_ops.py attn8 > gpurun_out/r3c_attn8_variants.log 2>&1 || { tail -20 gpurun_out/r3c_attn8_variants.log; exit 1; }
cat gpurun_out/r3c_attn8_variants.log
