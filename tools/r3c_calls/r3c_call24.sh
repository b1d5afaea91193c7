set -o pipefail
mkdir -p gpurun_out
B8="python bench.py --precision int8 --act-scales history --attention fp8 --no-cpu-baseline --no-fp8-leg"
for i in 1 2; do
timeout -k 10 200 $B8 > gpurun_out/r3c_spec2_off_$i.json 2>/dev/null || exit 2
TD_ATTN_TUNE=0x8000 timeout -k 10 200 $B8 > gpurun_out/r3c_spec2_on_$i.json 2>/dev/null || exit 3
done
python - <<'PY'
import json
for n in ("spec2_off_1", "spec2_on_1", "spec2_off_2", "spec2_on_2"):
    d = json.load(open(f"gpurun_out/r3c_{n}.json"))
    print(n, round(d["value"], 4), "one", round(d["one_image_in_flight"]["value"], 4), "attn ms", round(d["kernel_ms_per_image"]["attention"], 1))
PY
