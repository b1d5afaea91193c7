set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_attention_fp8_gpu.py tests/test_flux_full_depth_gpu.py tests/test_flux_engine_gpu.py tests/test_driver_gpu.py tests/test_torch_ops_gpu.py -x -q -m gpu > gpurun_out/r3c_tests27.log 2>&1 || { tail -60 gpurun_out/r3c_tests27.log; exit 1; }
tail -2 gpurun_out/r3c_tests27.log
timeout -k 10 500 python bench.py > gpurun_out/r3i_bench_line.json 2> gpurun_out/r3i_bench.err || exit 2
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3i_bench_line.json"))
print("bf16", round(d["value"], 4), "one", round(d["one_image_in_flight"]["value"], 4), "gemm frac", round(d["roofline"]["frac"], 4), "attn frac", round(d["attention_roofline"]["frac"], 4))
f = d["fp8"]
print("8-bit value", round(f["value"], 4), f["policy"], {k: (round(v["value"], 3), v["inside_1e-2_bar"]) for k, v in f["policies"].items()})
PY
