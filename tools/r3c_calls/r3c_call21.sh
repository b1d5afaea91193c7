set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_attention_fp8_gpu.py tests/test_flux_full_depth_gpu.py tests/test_flux_engine_gpu.py tests/test_driver_gpu.py -x -q -m gpu -s > gpurun_out/r3c_tests21.log 2>&1 || { tail -60 gpurun_out/r3c_tests21.log; exit 1; }
tail -2 gpurun_out/r3c_tests21.log; grep "history~oracle" gpurun_out/r3c_tests21.log | head -2; grep "attn8" gpurun_out/r3c_tests21.log | cut -c1-200
B8="python bench.py --precision int8 --act-scales history --attention fp8 --no-cpu-baseline --no-fp8-leg"
for i in 1 2; do
timeout -k 10 200 $B8 > gpurun_out/r3c_href_on_$i.json 2>/dev/null || exit 2
TD_ATTN8_NO_HREF=1 timeout -k 10 200 $B8 > gpurun_out/r3c_href_off_$i.json 2>/dev/null || exit 3
done
python - <<'PY'
import json
for n in ("href_on_1", "href_off_1", "href_on_2", "href_off_2"):
    d = json.load(open(f"gpurun_out/r3c_{n}.json"))
    print(n, round(d["value"], 4), "one", round(d["one_image_in_flight"]["value"], 4), "attn ms", round(d["kernel_ms_per_image"]["attention"], 1))
PY
