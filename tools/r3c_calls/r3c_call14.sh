set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_attention_fp8_gpu.py -x -q -m gpu > gpurun_out/r3c_tests14.log 2>&1 || { tail -50 gpurun_out/r3c_tests14.log; exit 1; }
tail -2 gpurun_out/r3c_tests14.log
timeout -k 10 300 python tools/bench_ops.py attn8 > gpurun_out/r3c_attn8_spec.log 2>&1 || { tail -20 gpurun_out/r3c_attn8_spec.log; exit 1; }
cat gpurun_out/r3c_attn8_spec.log
B8="python bench.py --precision int8 --act-scales history --attention fp8 --no-cpu-baseline --no-fp8-leg"
timeout -k 10 200 $B8 > gpurun_out/r3c_spec_off_1.json 2>/dev/null || exit 2
TD_ATTN_TUNE=0x8000 timeout -k 10 200 $B8 > gpurun_out/r3c_spec_on_1.json 2>/dev/null || exit 3
timeout -k 10 200 $B8 > gpurun_out/r3c_spec_off_2.json 2>/dev/null || exit 2
TD_ATTN_TUNE=0x8000 timeout -k 10 200 $B8 > gpurun_out/r3c_spec_on_2.json 2>/dev/null || exit 3
python - <<'PY'
import json
for n in ("spec_off_1", "spec_on_1", "spec_off_2", "spec_on_2"):
    d = json.load(open(f"gpurun_out/r3c_{n}.json"))
    print(n, round(d["value"], 4), "one", round(d["one_image_in_flight"]["value"], 4), "attn ms", round(d["kernel_ms_per_image"]["attention"], 1))
PY
