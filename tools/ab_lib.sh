for rep in 1 2; do
for L in lib lib_b; do
  TD_HIP_LIB=$PWD/thinkdiff-mlre_amd/$L/libthinkdiff_hip.so python bench.py --in-flight 1 --steps 3 --warmup 1 --no-cpu-baseline --no-fp8-leg > gpurun_out/r3_ab_$L.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r3_ab_$L.json')); print('$L', round(d['value'],4), round(d['attention_tflops'],1), {k:round(v,1) for k,v in d['kernel_ms_per_image'].items()})"
done; done
