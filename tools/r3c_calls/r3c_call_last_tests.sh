set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_flux_engine_gpu.py tests/test_attention_fp8_gpu.py -x -q -m gpu -s -k "score_bound or history or token_layout" > gpurun_out/r3c_last_tests.log 2>&1 || { tail -40 gpurun_out/r3c_last_tests.log; exit 1; }
tail -2 gpurun_out/r3c_last_tests.log; grep "norm weights x" gpurun_out/r3c_last_tests.log
