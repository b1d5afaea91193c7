set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_attention_gpu.py -x -q -m gpu > gpurun_out/r3c_tests19.log 2>&1 || { tail -40 gpurun_out/r3c_tests19.log; exit 1; }
tail -1 gpurun_out/r3c_tests19.log
B="python bench.py --no-cpu-baseline --no-fp8-leg"
for i in 1 2; do
timeout -k 10 200 $B > gpurun_out/r3c_rs_mfma_$i.json 2>/dev/null || exit 2
TD_ATTN_TUNE=0x400 timeout -k 10 200 $B > gpurun_out/r3c_rs_valu_$i.json 2>/dev/null || exit 3
done
python - <<'PY'
import json
for n in ("rs_mfma_1", "rs_valu_1", "rs_mfma_2", "rs_valu_2"):
    d = json.load(open(f"gpurun_out/r3c_{n}.json"))
    print(n, round(d["value"], 4), "one", round(d["one_image_in_flight"]["value"], 4), "attn ms", round(d["kernel_ms_per_image"]["attention"], 1), "attn us", round(d["attention_roofline"]["avg_launch_us"], 1))
PY
