set -o pipefail
mkdir -p gpurun_out
B="python bench.py --workload config5 --prompts 12 --precision int8 --act-scales history --attention fp8 --in-flight 2"
for i in 1 2; do
  timeout -k 10 200 $B > gpurun_out/r3c_rag_on_$i.json 2>/dev/null || exit 1
  TD_GEMM_NO_RAGGED=1 timeout -k 10 200 $B > gpurun_out/r3c_rag_off_$i.json 2>/dev/null || exit 2
done
B2="python bench.py --workload config5 --prompts 12 --precision bf16 --in-flight 2"
timeout -k 10 200 $B2 > gpurun_out/r3c_rag16_on.json 2>/dev/null || exit 3
TD_GEMM_NO_RAGGED=1 timeout -k 10 200 $B2 > gpurun_out/r3c_rag16_off.json 2>/dev/null || exit 4
python - <<'PY'
import json
for n in ("rag_on_1", "rag_off_1", "rag_on_2", "rag_off_2", "rag16_on", "rag16_off"):
    print(n, round(json.load(open(f"gpurun_out/r3c_{n}.json"))["value"], 4))
PY
