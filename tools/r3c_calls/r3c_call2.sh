# fused QK-norm/RoPE pack: tests, A/B against the two-pass form, images in flight, kernel stats + SQ counters of the int8 + fp8-attention image
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_attention_fp8_gpu.py -x -q -m gpu > gpurun_out/r3c_attn8_tests2.log 2>&1 || { tail -40 gpurun_out/r3c_attn8_tests2.log; exit 1; }
tail -2 gpurun_out/r3c_attn8_tests2.log
timeout -k 10 500 python -m pytest tests/test_attention_gpu.py -x -q -m gpu -k "ring or prescaled" > gpurun_out/r3c_attn_ring_tests.log 2>&1 || { tail -40 gpurun_out/r3c_attn_ring_tests.log; exit 1; }
tail -2 gpurun_out/r3c_attn_ring_tests.log
timeout -k 10 300 python tools/bench_ops.py attn > gpurun_out/r3c_attn_ring_ab.log 2>&1 || { tail -20 gpurun_out/r3c_attn_ring_ab.log; exit 1; }
cat gpurun_out/r3c_attn_ring_ab.log
B8="--precision int8 --act-scales history --attention fp8 --no-cpu-baseline --no-fp8-leg"
timeout -k 10 200 python bench.py $B8 > gpurun_out/r3c_i8a8_fused.json 2> gpurun_out/r3c_i8a8_fused.err || exit 2
TD_ATTN8_NO_FUSE=1 timeout -k 10 200 python bench.py $B8 > gpurun_out/r3c_i8a8_twopass.json 2> gpurun_out/r3c_i8a8_twopass.err || exit 3
timeout -k 10 200 python bench.py $B8 > gpurun_out/r3c_i8a8_fused_b.json 2> gpurun_out/r3c_i8a8_fused_b.err || exit 2
timeout -k 10 200 python bench.py $B8 --in-flight 3 > gpurun_out/r3c_i8a8_fused_g3.json 2> gpurun_out/r3c_i8a8_fused_g3.err || exit 4
python - <<'PY'
import json
for n in ("fused", "twopass", "fused_b", "fused_g3"):
    d = json.load(open(f"gpurun_out/r3c_i8a8_{n}.json"))
    print(n, round(d["value"], 4), "one at a time", round(d.get("one_image_in_flight", {}).get("value", 0), 4), {k: round(v, 1) for k, v in d.get("kernel_ms_per_image", {}).items()})
PY
B="python3 bench.py --precision int8 --act-scales history --attention fp8 --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-trace --no-fp8-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3c_stats -- $B > gpurun_out/r3c_stats.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/r3c_pmc_sq -- $B > gpurun_out/r3c_pmc_sq.log 2>&1 || exit 6
python tools/pmc_summary.py gpurun_out/r3c_pmc_sq td_ > gpurun_out/r3c_pmc_sq.txt
find gpurun_out/r3c_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3c_int8_attn8_kernel_stats.csv \;
rm -rf gpurun_out/r3c_pmc_sq gpurun_out/r3c_stats
head -12 gpurun_out/r3c_int8_attn8_kernel_stats.csv | cut -c1-200
echo done
