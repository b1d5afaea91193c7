# attention PV loop with inline-asm transposed reads (no compiler vmcnt(0) mid-tile) + ring forms: tests, A/B, headline
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_attention_gpu.py tests/test_flux_engine_gpu.py tests/test_qwen2_gpu.py tests/test_text_encoders_gpu.py tests/test_vision_towers_gpu.py -x -q -m gpu > gpurun_out/r3c_attn_tests3.log 2>&1 || { tail -40 gpurun_out/r3c_attn_tests3.log; exit 1; }
tail -2 gpurun_out/r3c_attn_tests3.log
timeout -k 10 300 python tools/bench_ops.py attn > gpurun_out/r3c_attn_ring_ab2.log 2>&1 || { tail -20 gpurun_out/r3c_attn_ring_ab2.log; exit 1; }
cat gpurun_out/r3c_attn_ring_ab2.log
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fp8-leg > gpurun_out/r3c_bf16_asmv.json 2> gpurun_out/r3c_bf16_asmv.err || exit 2
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3c_bf16_asmv.json"))
print("bf16", round(d["value"], 4), "one at a time", round(d["one_image_in_flight"]["value"], 4), {k: round(v, 1) for k, v in d["kernel_ms_per_image"].items()}, "attn us", round(d["attention_roofline"]["avg_launch_us"], 1), "gemm frac", round(d["roofline"]["frac"], 4))
PY
