set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/r3j_gpu_suite.log 2>&1 || { tail -40 gpurun_out/r3j_gpu_suite.log; exit 1; }
tail -2 gpurun_out/r3j_gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
