# In-flight count and fp8 policy under fully drawn synthetic weights (one GPU box, from the repo root)
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/r2_retune.log
for G in 1 2 3; do
  timeout -k 10 300 python bench.py --in-flight $G --steps 2 --warmup 1 --no-cpu-baseline --no-trace --no-fp8-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 in flight $G:', round(d['value'],4), 'images/s', round(d['ms_per_step'],1), 'ms/step')" >> gpurun_out/r2_retune.log || exit 1
done
for G in 1 2 3; do
  timeout -k 10 300 python bench.py --precision fp8 --in-flight $G --steps 2 --warmup 1 --no-cpu-baseline --no-trace 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp8 (T=193) in flight $G:', round(d['value'],4), 'images/s')" >> gpurun_out/r2_retune.log || exit 2
done
for A in 0 1; do
  TD_FLUX_INFLIGHT_ATTN=$A timeout -k 10 300 python bench.py --in-flight 2 --steps 2 --warmup 1 --no-cpu-baseline --no-trace --no-fp8-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 2 in flight, attention form $A (0 persistent, 1 plain grid):', round(d['value'],4))" >> gpurun_out/r2_retune.log || exit 3
done
timeout -k 10 600 python tools/fp8_policy_sweep.py 2>&1 | grep -v amdgpu.ids | grep "mask\|bf16" >> gpurun_out/r2_retune.log || exit 4
cat gpurun_out/r2_retune.log
