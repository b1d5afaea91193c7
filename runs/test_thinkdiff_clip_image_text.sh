# bash runs/test_thinkdiff_clip_image_text.sh 0 configs/test_thinkdiff_clip_image_text.yaml [--options run.synthetic=true ...]
gpu_id=$1
export HIP_VISIBLE_DEVICES=$gpu_id
cfg=$2
shift 2
python -m scripts.test.test_blip_vision_t5_decoder_flux_text --cfg-path $cfg "$@"
