#!/bin/bash
# Every GPU test file in its own fresh process, one after the other (order / first-call dependence check).
# Stops at the first file that times out; failures are collected.  Output: gpurun_out/isolated.log
mkdir -p gpurun_out
: > gpurun_out/isolated.log
fail=0
for f in tests/test_*_gpu.py; do
  echo "=== $f" >> gpurun_out/isolated.log
  timeout -k 10 600 python -m pytest "$f" -m gpu -q -p no:cacheprovider >> gpurun_out/isolated.log 2>&1
  rc=$?
  echo "=== $f rc=$rc" >> gpurun_out/isolated.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout in $f" >> gpurun_out/isolated.log; exit 1; fi
  if [ $rc -ne 0 ] && [ $rc -ne 5 ]; then fail=1; fi
done
grep "rc=" gpurun_out/isolated.log
exit $fail
