# rocprofv3 kernel stats + SQ counters of ONE image on the in-tolerance 8-bit path (int8 Linears under history scales + e4m3 attention), final source
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python3 bench.py --precision int8 --act-scales history --attention fp8 --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-trace --no-fp8-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3i_stats8 -- $B > gpurun_out/r3i_stats8.log 2>&1 || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/r3i_pmc_sq8 -- $B > gpurun_out/r3i_pmc_sq8.log 2>&1 || exit 6
python tools/pmc_summary.py gpurun_out/r3i_pmc_sq8 td_ > gpurun_out/r3i_pmc_sq_int8_attn8.txt
find gpurun_out/r3i_stats8 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r3i_int8_attn8_kernel_stats.csv \;
rm -rf gpurun_out/r3i_pmc_sq8 gpurun_out/r3i_stats8
head -8 gpurun_out/r3i_int8_attn8_kernel_stats.csv | cut -c1-200
grep -A9 "attn_fwd_d128_fp8" gpurun_out/r3i_pmc_sq_int8_attn8.txt | head -10
