"""BASELINE config 3 end to end on synthetic full-size weights: ThinkDiff-LVLM driver (Qwen2-VL-7B-shaped ViT + decoder with 128
sampled tokens -> aligner -> FLUX.1-dev-shaped 1024^2 28 steps -> VAE -> PNG).  Prints seconds per request, warm."""
import os, sys, time
import torch
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.test import test_mllama_t5_decoder_flux as drv

tmp = os.environ.get("TMPDIR", "/tmp")
img = os.path.join(tmp, "dot_image.jpeg")
Image.new("RGB", (640, 480), (30, 90, 200)).save(img)
out = os.path.join(tmp, "lvlm_out")
argv = ["--cfg-path", os.path.join(ROOT, "tests", "golden", "thinkdiff_lvlm_driver_keys.yaml"), "--options", "run.synthetic=true",
        "run.distributed=false", f"run.img_urls=[{img}]", f"run.output_dir={out}", "model.ckpt="] + sys.argv[1:]
args = drv.parse_args(argv)
cfg = drv.Config(args)
drv.setup_seeds(42)
d = drv.LvlmFluxDriver(cfg)
torch.cuda.synchronize()
for rnd in range(2):
    t0 = time.perf_counter()
    sample = {"images": [[Image.open(img).convert("RGB")]], "answers": list(drv.DEFAULT_ANSWERS)}
    emb, txt = d.model.get_embed(sample, embedding_type="output_embed", max_new_tokens=128)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    w = d.run()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"round {rnd}: get_embed (ViT + prefill + 128 sampled tokens + aligner) {t1-t0:.3f} s; full driver run (get_embed + CLIP pooled + FLUX + VAE + PNG) {t2-t1:.3f} s; tokens {emb[0].shape}", flush=True)
