"""BASELINE config 4 end to end on one GPU: scripts/generate_embedding_webdataset over synthetic WebDataset shards (500x375 JPEGs
with natural-image-like content, Qwen2-VL-2B-shaped synthetic weights, 64 generated tokens) -> samples/s of the whole job
(tar read + JPEG decode + model + JPEG encode + torch.save + tar write), beside the model-forward-only figure of
tools/bench_precompute.py.   usage: bench_precompute_job.py [n_samples=1024] [loader batch=512]"""
import os, shutil, sys, time
import numpy as np
import torch
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))
from scripts import generate_embedding_webdataset as job
from thinkdiff.datasets import wds_io
from thinkdiff.tasks import image_text_process_data as task_mod

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 512
root = "/tmp/td_job_bench"
shutil.rmtree(root, ignore_errors=True)
os.makedirs(root)
rng = np.random.default_rng(0)
base = rng.integers(0, 256, (24, 32, 3), dtype=np.uint8)
shards, k = [], 0
per = 256
for s in range((n + per - 1) // per):
    path = os.path.join(root, f"in-{s:05d}.tar")
    w = wds_io.TarWriter(path)
    for _ in range(min(per, n - k)):
        img = Image.fromarray(np.roll(base, k, axis=1)).resize((500, 375), Image.BICUBIC)      # smooth content: realistic JPEG sizes
        w.write({"__key__": f"sample{k:06d}", "jpg": img, "json": {"caption": f"caption {k}"}})
        k += 1
    w.close()
    shards.append({"url": path, "nsamples": min(per, n - s * per)})
idx = os.path.join(root, "wids_shards.json")
wds_io.write_wids_index(idx, shards, name="bench")
argv = ["--cfg-path", os.path.join(ROOT, "tests", "golden", "qwen2_vl_embed_keys.yaml"), "--options", "run.synthetic=true",
        "run.synthetic_max_image_tokens=320", f"datasets.cc_sbu_mllama_vllm_process_wids.build_info.storage={idx}",
        f"datasets.cc_sbu_mllama_vllm_process_wids.batch_size={bs}", f"run.output_shard_path=[{root}/out,'%06d.tar',0]",
        "model.vllm_config.max_model_len=2048", f"model.vllm_config.max_tokens={os.environ.get('TD_JOB_TOKENS', '64')}", f"model.vllm_config.min_tokens={os.environ.get('TD_JOB_TOKENS', '64')}", "model.vllm_config.ignore_eos=true",
        f"model.vllm_config.max_num_seqs={os.environ.get('TD_JOB_SEQS', '256')}", f"model.vllm_config.max_num_batched_tokens=60000",
        "model.text_config={hidden_size: 1536, num_hidden_layers: 28, num_attention_heads: 12, num_key_value_heads: 2, intermediate_size: 8960, vocab_size: 151936, tie_word_embeddings: true}"]
if os.environ.get("TD_JOB_STOP_EVERY"):      # outputs of different lengths (synthetic weights never sample one particular EOS id): every k-th token id ends a sequence
    k = int(os.environ["TD_JOB_STOP_EVERY"])
    argv[-1:-1] = ["model.vllm_config.min_tokens=1", "model.vllm_config.stop_token_ids=[" + ",".join(str(i) for i in range(0, 151936, k)) + "]"]
t = {}
orig = task_mod.ImageTextProcessDataTask.train_epoch
def timed(self, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = orig(self, *a, **k)
    torch.cuda.synchronize(); t["epoch"] = time.perf_counter() - t0
    return r
task_mod.ImageTextProcessDataTask.train_epoch = timed
if os.environ.get("TD_PROFILE"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    res = job.main(argv)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
else:
    res = job.main(argv)
stats = res[0] if isinstance(res, list) else res
print(f"precompute job: {stats['samples']} samples in {t['epoch']:.2f} s = {stats['samples'] / t['epoch']:.1f} samples/s end to end "
      f"(loader batch {bs}, {len(stats['shards'])} output shards, {sum(os.path.getsize(s['url']) for s in stats['shards']) / 1e6:.0f} MB written)")
