# bash runs/test_thinkdiff_clip_two_images.sh 0 configs/test_thinkdiff_clip_two_images.yaml [--options run.synthetic=true run.flux_precision=fp8 ...]
# (reference runs/test_thinkdiff_clip_two_images.sh; several GPUs, e.g. "0,1,2,3,4,5,6,7": one process per GPU, and with
#  run.shard_prompts=true the (composition, prompt) jobs are split over the ranks instead of replicated with seed + rank)
gpu_id=$1
export HIP_VISIBLE_DEVICES=$gpu_id
gpu_num=$(echo $HIP_VISIBLE_DEVICES | tr ',' '\n' | wc -l)
cfg=$2
shift 2
if [ "$gpu_num" -gt 1 ]; then
  torchrun --nproc-per-node $gpu_num --master-addr 127.0.0.1 --master-port 10000 -m scripts.test.test_blip_vision_t5_decoder_flux --cfg-path $cfg "$@"
else
  python -m scripts.test.test_blip_vision_t5_decoder_flux --cfg-path $cfg "$@"
fi
