set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gemm_gpu.py tests/test_int8_gpu.py tests/test_fp8_gpu.py tests/test_flux_engine_gpu.py tests/test_vae_gpu.py -x -q -m gpu > gpurun_out/r3c_tests9.log 2>&1 || { tail -50 gpurun_out/r3c_tests9.log; exit 1; }
tail -2 gpurun_out/r3c_tests9.log
timeout -k 10 300 python tools/bench_ragged.py > gpurun_out/r3c_gemm_ragged_ab.log 2>&1 || { tail gpurun_out/r3c_gemm_ragged_ab.log; exit 2; }
cat gpurun_out/r3c_gemm_ragged_ab.log
