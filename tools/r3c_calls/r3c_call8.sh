set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_torch_ops_gpu.py tests/test_vae_gpu.py tests/test_flux_engine_gpu.py tests/test_driver_gpu.py tests/test_attention_fp8_gpu.py -x -q -m gpu > gpurun_out/r3c_tests8.log 2>&1 || { tail -50 gpurun_out/r3c_tests8.log; exit 1; }
tail -2 gpurun_out/r3c_tests8.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp8-leg --steps 2 > gpurun_out/r3c_bf16_ops.json 2> gpurun_out/r3c_bf16_ops.err || { tail gpurun_out/r3c_bf16_ops.err; exit 2; }
python -c "import json; d=json.load(open('gpurun_out/r3c_bf16_ops.json')); print('bf16 via torch.ops', round(d['value'],4), round(d['one_image_in_flight']['value'],4))"
