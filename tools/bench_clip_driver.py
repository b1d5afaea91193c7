"""BASELINE config 2 / 5 through the real driver on full-size synthetic weights: ThinkDiff-CLIP image(+image)+text -> PNG.
EVA-ViT-g + aligner + T5-XXL + CLIP-L + FLUX.1-dev shape + VAE, 1024^2, 28 steps; groups of `images_in_flight`."""
import os, sys, time, json
import torch
from PIL import Image
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scripts.test import test_blip_vision_t5_decoder_flux_text as drv

tmp = os.environ.get("TMPDIR", "/tmp")
imgs = []
for k in range(6):
    p = os.path.join(tmp, f"clipdrv_{k}.jpg")
    Image.new("RGB", (640, 480), (40 * k % 255, 90, 200 - 20 * k)).save(p)
    imgs.append(p)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
two = len(sys.argv) > 2 and sys.argv[2] == "two"
urls = "[" + ",".join(f"[{imgs[i]},{imgs[(i+1)%6]}]" if two else imgs[i] for i in range(6)) + "]"
out = os.path.join(tmp, f"clipdrv_out_{prec}_{int(two)}")
argv = ["--cfg-path", os.path.join(ROOT, "tests", "golden", "thinkdiff_clip_driver_keys.yaml"), "--options", "run.synthetic=true",
        f"run.img_urls={urls}", "run.questions=[a red apple on a wooden table]", "run.questions_names=[apple]", "run.prompt_json=",
        "run.use_image_name_and_prompt_as_output_name=false", f"run.output_dir={out}", f"run.flux_precision={prec}", "model.ckpt="]
args = drv.parse_args(argv)
cfg = drv.Config(args)
drv.setup_seeds(42)
d = drv.ClipFluxDriver(cfg)
torch.cuda.synchronize()
t0 = time.perf_counter()
w = d.run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{prec} {'two-image' if two else 'single-image'} driver: {len(w)} PNGs in {dt:.2f} s = {len(w)/dt:.3f} images/s (includes first-call warm-up)")
import shutil; shutil.rmtree(out)
t0 = time.perf_counter(); w = d.run(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"{prec} {'two-image' if two else 'single-image'} driver, warm: {len(w)} PNGs in {dt:.2f} s = {len(w)/dt:.3f} images/s")
