# Round-3 final measurement set (after the 8-bit attention / attention wait fixes) (run on the GPU box from the repo root): headline line, kernel stats, HBM traffic passes, SQ counters.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r3h_bench_line.json 2> gpurun_out/r3h_bench.err || exit 1
B="python3 bench.py --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-trace --no-fp8-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3h_stats -- $B > gpurun_out/r3h_stats.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3h_pmc_fetch -- $B > gpurun_out/r3h_pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3h_pmc_write -- $B > gpurun_out/r3h_pmc_write.log 2>&1 || exit 4
python tools/pmc_traffic.py gpurun_out/r3h_pmc_fetch gpurun_out/r3h_pmc_write gpurun_out/r3h_hbm_traffic.json > gpurun_out/r3h_hbm_traffic.txt || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/r3h_pmc_sq -- $B > gpurun_out/r3h_pmc_sq.log 2>&1 || exit 6
python tools/pmc_summary.py gpurun_out/r3h_pmc_sq td_ > gpurun_out/r3h_pmc_sq.txt
rm -rf gpurun_out/r3h_pmc_fetch gpurun_out/r3h_pmc_write gpurun_out/r3h_pmc_sq
echo done
TD_BENCH_FORCE_DIST=1 timeout -k 10 200 python bench.py --workload config5 --prompts 8 --precision int8 --act-scales history --attention fp8 > gpurun_out/r3h_config5_int8_attn8_8prompts.json 2> gpurun_out/r3h_config5.err || exit 7
cat gpurun_out/r3h_config5_int8_attn8_8prompts.json | cut -c1-300
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3h_bench_line.json"))
print("bf16", round(d["value"], 4), "one", round(d["one_image_in_flight"]["value"], 4), "gemm frac", round(d["roofline"]["frac"], 4), "attn frac", round(d["attention_roofline"]["frac"], 4))
f = d["fp8"]
print("8-bit value", round(f["value"], 4), f["policy"], {k: (round(v["value"], 3), v["inside_1e-2_bar"]) for k, v in f["policies"].items()})
print("cpu", d["cpu_baseline"]["value"])
PY
