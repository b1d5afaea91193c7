"""ThinkDiff-LVLM text-only prompt -> aligner tokens -> FLUX driver on the MI355X path.

Mirror of the reference driver scripts/test/test_mllama_t5_decoder_flux_multi_image_input_embed.py:135-291: the user turn is
the bare prompt (no image parts, :190-212), templated with `add_vision_id=True`, `get_embed({"prompt": ...},
need_process=False, max_new_tokens=128)`; the aligner output is cut or zero-padded to `run.max_tokens` rows when that key is
set (:258-264), pooled = CLIP(""), seeds set again before the FLUX call (:274), 512 x 512, 28 steps, guidance 3.5, output
`{image_name}_edit_4_flux_{embedding_type}_{i}_{ckpt_id}.png` (:289-291).

The reference hard-codes the prompt and name (its last assignments, :187-188, are the defaults); `run.prompt` and
`run.image_names` replace them.

    python -m scripts.test.test_mllama_t5_decoder_flux_multi_image_input_embed --cfg-path <lvlm yaml> \
        [--options run.synthetic=true run.prompt="a photo of a dog" run.image_names=[dog] run.max_tokens=128]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))

from scripts.test.test_mllama_t5_decoder_flux import setup_seeds  # noqa: E402
from scripts.test.test_mllama_t5_decoder_flux_multi_image import SYSTEM_PROMPT, LvlmMultiImageFluxDriver, main as _main  # noqa: E402
from thinkdiff.common.dist_utils import get_rank  # noqa: E402


def fit_tokens(tokens, max_length):
    """[1, T, C] -> [1, max_length, C]: cut, or pad with zero rows (reference :258-264); None leaves T alone."""
    if max_length is None or tokens.shape[1] == max_length:
        return tokens
    if tokens.shape[1] > max_length:
        return tokens[:, :max_length]
    pad = torch.zeros((tokens.shape[0], max_length - tokens.shape[1], tokens.shape[2]), dtype=tokens.dtype, device=tokens.device)
    return torch.cat([tokens, pad], dim=1)


class LvlmTextPromptFluxDriver(LvlmMultiImageFluxDriver):
    PROMPT = "a photo of a pink skateboard"       # reference :187
    IMAGE_NAMES = ["skateboard"]                  # reference :188

    def run(self):
        run = self.cfg.run_cfg
        out_dir = run["output_dir"]
        os.makedirs(out_dir, exist_ok=True)
        seed = run.seed + get_rank()
        ckpt_id = os.path.basename(self.cfg.model_cfg["ckpt"] or "")
        question = run.get("prompt", None) if run.get("prompt", None) is not None else self.PROMPT
        image_names = list(run.get("image_names", None) or self.IMAGE_NAMES)
        messages = [{"role": "system", "content": SYSTEM_PROMPT}, {"role": "user", "content": [{"type": "text", "text": question}]}]
        prompt = self.model.module.mllama_processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, add_vision_id=True)
        print(prompt)
        embedding_type = self.cfg.model_cfg.get("embedding_type", "output_embed")
        with torch.no_grad():
            language_model_inputs, generated = self.model.module.get_embed({"prompt": prompt}, embedding_type=embedding_type, max_new_tokens=128, need_process=False)
        for i, text in enumerate(generated):
            print(language_model_inputs[i].shape)
            print(text)
        written = []
        for img_i in range(1):                                    # reference :251
            pe = fit_tokens(language_model_inputs[img_i].unsqueeze(0), run.get("max_tokens", None)).to(torch.bfloat16)
            pooled = self.pooled_empty_prompt().to(torch.bfloat16)
            setup_seeds(seed)
            images = self.pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, num_images_per_prompt=1, height=self.HEIGHT, width=self.WIDTH,
                               num_inference_steps=28, guidance_scale=3.5).images
            for image_i, image in enumerate(images):
                path = f"{out_dir}/{image_names[img_i]}_edit_4_flux_{embedding_type}_{image_i}_{ckpt_id}.png"
                image.save(path, format="PNG", compress_level=1)
                print(f"Saved image to {path}")
                written.append(path)
        return written


def main(argv=None, driver_cls=LvlmTextPromptFluxDriver):
    return _main(argv, driver_cls)


if __name__ == "__main__":
    main()
