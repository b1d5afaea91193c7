set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_attention_fp8_gpu.py tests/test_flux_engine_gpu.py tests/test_driver_gpu.py tests/test_int8_gpu.py tests/test_torch_ops_gpu.py -x -q -m gpu > gpurun_out/r3c_tests12.log 2>&1 || { tail -50 gpurun_out/r3c_tests12.log; exit 1; }
tail -2 gpurun_out/r3c_tests12.log
timeout -k 10 300 python bench.py --workload config5 --prompts 8 > gpurun_out/r3c_config5_default.json 2> gpurun_out/r3c_config5_default.err || { tail gpurun_out/r3c_config5_default.err; exit 2; }
cut -c1-900 gpurun_out/r3c_config5_default.json
