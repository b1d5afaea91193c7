# Decode step of the Qwen2-VL engine under rocprofv3 (kernel durations vs the step's wall time), graph replay and eager.  Run from the repo root on the GPU box.
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
B="python3 tools/bench_decode_batch.py ${1:-2B} ${2:-1}"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f_dec_graph -- $B > gpurun_out/r4f_dec_graph.log 2>&1 || exit 2
TD_QWEN2_NO_GRAPH=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4f_dec_eager -- $B > gpurun_out/r4f_dec_eager.log 2>&1 || exit 3
for d in graph eager; do
  f=$(find gpurun_out/r4f_dec_$d -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/r4f_decode_${d}_kernel_stats.csv; head -16 "$f" | cut -c1-160; fi
  grep "decode B" gpurun_out/r4f_dec_$d.log
  rm -rf gpurun_out/r4f_dec_$d
done
echo done
