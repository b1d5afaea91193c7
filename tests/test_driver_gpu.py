"""The ThinkDiff-CLIP driver end to end on the GPU (BASELINE config 1 in miniature: 4-step 256x256, synthetic
assets): config surface, prompt json, naming rule, aligner -> FLUX -> VAE -> PNG, skip-if-exists."""
import json
import os

import pytest
from PIL import Image

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_clip_image_text_driver_writes_png(hip, tmp_path):
    from scripts.test import test_blip_vision_t5_decoder_flux_text as drv
    img = tmp_path / "IP_Adapter_vermeer.jpg"
    Image.new("RGB", (300, 200), (120, 80, 40)).save(img)
    pj = tmp_path / "prompts.json"
    pj.write_text(json.dumps({"IP_Adapter_vermeer": "The girl holds a board showing 'Think DIFFERENT.'."}))
    out = tmp_path / "out"
    argv = ["--cfg-path", os.path.join(HERE, "golden", "thinkdiff_clip_driver_keys.yaml"), "--options",
            "run.synthetic=true", "run.synthetic_tiny=true", "run.flux_height=256", "run.flux_width=256",
            "run.flux_num_inference_steps=4", f"run.output_dir={out}", f"run.img_urls=[{img}]", f"run.prompt_json={pj}",
            "model.ckpt="]
    written = drv.main(argv)
    assert len(written) == 1
    name = os.path.basename(written[0])
    assert name == "IP_Adapter_vermeer_The_girl_holds_a_board_showing_Think_DIFFERENT.png"   # use_image_name_and_prompt_as_output_name
    im = Image.open(written[0])
    assert im.size == (256, 256) and im.mode == "RGB"
    assert drv.main(argv) == []          # second run: "Image already exists", nothing rendered
