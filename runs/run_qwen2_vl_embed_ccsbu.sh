# bash runs/run_qwen2_vl_embed_ccsbu.sh 0,1,2,3,4,5,6,7 [--options run.synthetic=true ...]
# (reference runs/run_qwen2_vl_embed_ccsbu.sh; here one process per listed GPU, the shard list is split by rank)
gpu_id=$1
export HIP_VISIBLE_DEVICES=$gpu_id
gpu_num=$(echo $HIP_VISIBLE_DEVICES | tr ',' '\n' | wc -l)
shift 1
torchrun --nproc-per-node $gpu_num --master-addr 127.0.0.1 --master-port 9997 -m scripts.generate_embedding_webdataset --cfg-path configs/qwen2_vl_embed_ccsbu.yaml "$@"
