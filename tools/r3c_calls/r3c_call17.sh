set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_attention_gpu.py tests/test_flux_engine_gpu.py tests/test_flux_full_depth_gpu.py tests/test_driver_gpu.py -x -q -m gpu -s > gpurun_out/r3c_tests17.log 2>&1 || { tail -60 gpurun_out/r3c_tests17.log; exit 1; }
tail -2 gpurun_out/r3c_tests17.log; grep "full depth\]" gpurun_out/r3c_tests17.log | cut -c1-260
B="python bench.py --no-cpu-baseline --no-fp8-leg"
for i in 1 2; do
timeout -k 10 200 $B > gpurun_out/r3c_bound_on_$i.json 2>/dev/null || exit 2
TD_ATTN_NO_BOUND=1 timeout -k 10 200 $B > gpurun_out/r3c_bound_off_$i.json 2>/dev/null || exit 3
done
python - <<'PY'
import json
for n in ("bound_on_1", "bound_off_1", "bound_on_2", "bound_off_2"):
    d = json.load(open(f"gpurun_out/r3c_{n}.json"))
    print(n, round(d["value"], 4), "one", round(d["one_image_in_flight"]["value"], 4), "attn ms", round(d["kernel_ms_per_image"]["attention"], 1), "attn us", round(d["attention_roofline"]["avg_launch_us"], 1), "frac", round(d["attention_roofline"]["frac"], 4))
PY
