# fp8-attention tests, full-depth parity with the attn8 policies, int8 bench with / without fp8 attention
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_attention_fp8_gpu.py -x -q -m gpu > gpurun_out/r3c_attn8_tests.log 2>&1 || { tail -30 gpurun_out/r3c_attn8_tests.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_flux_full_depth_gpu.py -x -q -m gpu -s > gpurun_out/r3c_full_depth.log 2>&1 || { tail -30 gpurun_out/r3c_full_depth.log; exit 2; }
timeout -k 10 200 python bench.py --precision int8 --act-scales history --attention fp8 --no-cpu-baseline --no-fp8-leg > gpurun_out/r3c_int8_attn8.json 2> gpurun_out/r3c_int8_attn8.err || { tail -20 gpurun_out/r3c_int8_attn8.err; exit 3; }
timeout -k 10 200 python bench.py --precision int8 --act-scales history --no-cpu-baseline --no-fp8-leg > gpurun_out/r3c_int8.json 2> gpurun_out/r3c_int8.err || exit 4
tail -3 gpurun_out/r3c_attn8_tests.log gpurun_out/r3c_full_depth.log
cat gpurun_out/r3c_int8_attn8.json gpurun_out/r3c_int8.json
