# bash runs/test_thinkdiff_lvlm.sh 0 <lvlm yaml> [--options run.synthetic=true run.image_paths=[a.jpg,b.jpg] ...]
# (reference runs/test_thinkdiff_lvlm.sh: the interleaved words + pictures driver, one process per listed GPU, seed + rank;
#  the reference names a config that is not in its tree, so the config path is an argument here)
gpu_id=$1
export HIP_VISIBLE_DEVICES=$gpu_id
gpu_num=$(echo $HIP_VISIBLE_DEVICES | tr ',' '\n' | wc -l)
cfg=$2
shift 2
torchrun --nproc-per-node $gpu_num --master-addr 127.0.0.1 --master-port 9999 -m scripts.test.test_mllama_t5_decoder_flux_multi_image --cfg-path $cfg "$@"
