# A/B of an environment switch inside one gpurun call: bash tools/ab_env.sh VAR "v1 v2" [reps]
VAR=$1; VALS=$2; REPS=${3:-2}
for rep in $(seq $REPS); do for v in $VALS; do
  env $VAR=$v python bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp8-leg > gpurun_out/ab_${VAR}_$v.json 2>gpurun_out/ab_${VAR}_$v.err
  python -c "
import json; d=json.load(open('gpurun_out/ab_${VAR}_$v.json')); print('$VAR=$v', round(d['value'],4), 'roofline', round(d['roofline']['frac'],4), {k:round(x,1) for k,x in d['kernel_ms_per_image'].items()})"
done; done
