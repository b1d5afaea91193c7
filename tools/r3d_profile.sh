# Round-3 final measurement set (after the 8-bit attention / attention wait fixes) (run on the GPU box from the repo root): headline line, kernel stats, HBM traffic passes, SQ counters.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 500 python bench.py > gpurun_out/r3d_bench_line.json 2> gpurun_out/r3d_bench.err || exit 1
B="python3 bench.py --in-flight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-trace --no-fp8-leg"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3d_stats -- $B > gpurun_out/r3d_stats.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r3d_pmc_fetch -- $B > gpurun_out/r3d_pmc_fetch.log 2>&1 || exit 3
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r3d_pmc_write -- $B > gpurun_out/r3d_pmc_write.log 2>&1 || exit 4
python tools/pmc_traffic.py gpurun_out/r3d_pmc_fetch gpurun_out/r3d_pmc_write gpurun_out/r3d_hbm_traffic.json > gpurun_out/r3d_hbm_traffic.txt || exit 5
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d gpurun_out/r3d_pmc_sq -- $B > gpurun_out/r3d_pmc_sq.log 2>&1 || exit 6
python tools/pmc_summary.py gpurun_out/r3d_pmc_sq td_ > gpurun_out/r3d_pmc_sq.txt
rm -rf gpurun_out/r3d_pmc_fetch gpurun_out/r3d_pmc_write gpurun_out/r3d_pmc_sq
echo done
