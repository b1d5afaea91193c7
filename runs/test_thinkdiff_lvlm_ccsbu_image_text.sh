# bash runs/test_thinkdiff_lvlm_ccsbu_image_text.sh 0,1 [--options run.synthetic=true ...]
# (reference runs/test_thinkdiff_lvlm_ccsbu_image_text.sh: one process per listed GPU, each renders with seed + rank)
gpu_id=$1
export HIP_VISIBLE_DEVICES=$gpu_id
gpu_num=$(echo $HIP_VISIBLE_DEVICES | tr ',' '\n' | wc -l)
shift 1
torchrun --nproc-per-node $gpu_num --master-addr 127.0.0.1 --master-port 9998 -m scripts.test.test_mllama_t5_decoder_flux --cfg-path configs/test_thinkdiff_lvlm_ccsbu_image_text.yaml "$@"
