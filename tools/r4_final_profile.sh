# Round-4 final measurement set (run on the GPU box from the repo root): the headline line, rocprof kernel stats, HBM traffic passes and SQ counters
# of the bf16 image; kernel stats + SQ counters of the 8-bit image (int8 Linears: smoothing + history scales, e4m3 attention); the config-5 rehearsal.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
R=${1:-r4}
timeout -k 10 600 python bench.py < /dev/null > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench.err || exit 1
B="python3 bench.py --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-trace --no-fp8-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats -- $B < /dev/null > gpurun_out/${R}_stats.log 2>&1 || exit 2
find gpurun_out/${R}_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R}_bench_kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/${R}_pmc_fetch -- $B < /dev/null > gpurun_out/${R}_pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/${R}_pmc_write -- $B < /dev/null > gpurun_out/${R}_pmc_write.log 2>&1 || exit 4
python tools/pmc_traffic.py gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write gpurun_out/${R}_hbm_traffic.json > gpurun_out/${R}_hbm_traffic.txt || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/${R}_pmc_sq -- $B < /dev/null > gpurun_out/${R}_pmc_sq.log 2>&1 || exit 6
python tools/pmc_summary.py gpurun_out/${R}_pmc_sq td_ > gpurun_out/${R}_pmc_sq.txt
rm -rf gpurun_out/${R}_pmc_fetch gpurun_out/${R}_pmc_write gpurun_out/${R}_pmc_sq gpurun_out/${R}_stats
echo "bf16 set done"
B8="python3 bench.py --precision int8 --act-scales history --smoothing on --attention fp8 --in-flight 1 --steps 1 --warmup 1 --no-cpu-baseline --no-trace --no-fp8-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${R}_stats8 -- $B8 < /dev/null > gpurun_out/${R}_stats8.log 2>&1 || exit 7
find gpurun_out/${R}_stats8 -name "*kernel_stats.csv" -exec cp {} gpurun_out/${R}_int8_smooth_attn8_kernel_stats.csv \;
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/${R}_pmc_sq8 -- $B8 < /dev/null > gpurun_out/${R}_pmc_sq8.log 2>&1 || exit 8
python tools/pmc_summary.py gpurun_out/${R}_pmc_sq8 td_ > gpurun_out/${R}_pmc_sq_int8_smooth_attn8.txt
rm -rf gpurun_out/${R}_pmc_sq8 gpurun_out/${R}_stats8
TD_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --workload config5 --prompts 8 < /dev/null > gpurun_out/${R}_config5_8prompts.json 2> gpurun_out/${R}_config5.err || exit 9
head -8 gpurun_out/${R}_bench_kernel_stats.csv | cut -c1-200
head -8 gpurun_out/${R}_int8_smooth_attn8_kernel_stats.csv | cut -c1-200
python - <<PY
import json
d = json.load(open("gpurun_out/${R}_bench_line.json"))
print("bf16", round(d["value"], 4), "one", round(d["one_image_in_flight"]["value"], 4), "gemm frac", round(d["roofline"]["frac"], 4), "attn frac", round(d["attention_roofline"]["frac"], 4))
f = d["fp8"]
print("8-bit value", round(f["value"], 4), f["policy"], {k: (round(v["value"], 3), v["inside_1e-2_bar"]) for k, v in f["policies"].items()})
print("cpu", d["cpu_baseline"]["value"])
c = json.load(open("gpurun_out/${R}_config5_8prompts.json"))
print("config5 8 prompts", round(c["value"], 4), c["config"]["smoothing"], c["parity"]["inside_1e-2_bar_on_every_fixture"])
PY
echo done
