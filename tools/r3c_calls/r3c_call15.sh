set -o pipefail
mkdir -p gpurun_out
B8="python bench.py --precision int8 --act-scales history --attention fp8 --no-cpu-baseline --no-fp8-leg"
for i in 1 2; do
timeout -k 10 200 $B8 > gpurun_out/r3c_h7_$i.json 2>/dev/null || exit 2
TD_ATTN_TUNE=0x4000 timeout -k 10 200 $B8 > gpurun_out/r3c_h5_$i.json 2>/dev/null || exit 3
TD_ATTN_TUNE=0x8000 timeout -k 10 200 $B8 > gpurun_out/r3c_h3_$i.json 2>/dev/null || exit 3
done
python - <<'PY'
import json
for n in ("h7_1", "h5_1", "h3_1", "h7_2", "h5_2", "h3_2"):
    d = json.load(open(f"gpurun_out/r3c_{n}.json"))
    print(n, round(d["value"], 4), "one", round(d["one_image_in_flight"]["value"], 4), "attn ms", round(d["kernel_ms_per_image"]["attention"], 1))
PY
for h in 0x4000 0x8000; do
TD_ATTN_TUNE=$h timeout -k 10 600 python -m pytest tests/test_flux_full_depth_gpu.py -x -q -m gpu -s -k "fp8_policies" > gpurun_out/r3c_full_depth_$h.log 2>&1 || { tail -30 gpurun_out/r3c_full_depth_$h.log; }
echo "TD_ATTN_TUNE=$h"; grep "attn8\|passed\|failed" gpurun_out/r3c_full_depth_$h.log | cut -c1-200
done
