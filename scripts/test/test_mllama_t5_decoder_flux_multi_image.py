"""ThinkDiff-LVLM interleaved words + pictures -> FLUX driver on the MI355X path ("what is the next picture").

Mirror of the reference driver scripts/test/test_mllama_t5_decoder_flux_multi_image.py:80-282 (run by
runs/test_thinkdiff_lvlm.sh): one chat request -- system turn, user turn = [question, (text_i, image_i)..., last text] --
templated with `add_vision_id=True` ("Picture N: " in front of every image), images through `process_vision_info`,
`get_embed(sample, need_process=False, max_new_tokens=128)`, the aligner output conditions FLUX in place of the T5
embeddings, pooled = CLIP(""), the seeds are set again right before the FLUX call (:252), 512 x 512, 28 steps,
guidance 3.5, output `{image_name}_output_embed_edit_4_flux_{i}_{ckpt_id}.png` with PNG compress_level=1 (:267-268).

The reference hard-codes its inputs (:179-196); they are the defaults here and can be replaced from the config:
`run.image_paths`, `run.texts` (one per image + the trailing one), `run.image_names`, `run.question`.

    python -m scripts.test.test_mllama_t5_decoder_flux_multi_image --cfg-path <lvlm yaml> \
        [--options run.synthetic=true run.image_paths=[a.jpg,b.jpg] ...]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "thinkdiff-mlre_amd"))

from scripts.test.test_mllama_t5_decoder_flux import LvlmFluxDriver, parse_args, setup_seeds  # noqa: E402
from thinkdiff.common.config import Config  # noqa: E402
from thinkdiff.common.dist_utils import get_rank, init_distributed_mode  # noqa: E402
from thinkdiff.runners import dp_inference as dp  # noqa: E402

SYSTEM_PROMPT = "You are a helpful assistant."


def build_messages(question, image_paths, texts, question_in_chat=True, max_pixels=None):
    """Reference ..._multi_image.py:198-219 (question first) / ..._multi_image_input.py:253-277 (no question part, every
    image capped at `max_pixels`)."""
    placeholders = []
    for i, path in enumerate(image_paths):
        placeholders.append({"type": "text", "text": texts[i]})
        placeholders.append({"type": "image", "image": path} if max_pixels is None else {"type": "image", "image": path, "max_pixels": max_pixels})
    placeholders.append({"type": "text", "text": texts[-1]})
    content = ([{"type": "text", "text": question}] if question_in_chat else []) + placeholders
    return [{"role": "system", "content": SYSTEM_PROMPT}, {"role": "user", "content": content}]


class LvlmMultiImageFluxDriver(LvlmFluxDriver):
    # reference ..._multi_image.py:139, :179-196
    QUESTION = ("I give you several words and pictures. First, please analyse what the next picture is. Then give me a detailed "
                "diffusion prompt to describe the next picture. Please only provide me the detailed prompt and start the answer "
                "with 'Create an image'.\n\n")
    IMAGE_PATHS = ["/root/dataset/minigpt4/cobsat/datasets/color_car/white_car.jpg",
                   "/root/dataset/minigpt4/cobsat/datasets/color_car/blue_car.jpg"]
    TEXTS = ["Word 1: white, ", "\n\nWord 2: blue, ", "\n\nWord 3: red, "]
    IMAGE_NAMES = ["car_white_blue_red"]
    QUESTION_IN_CHAT = True           # the question is the first part of the user turn
    MAX_PIXELS = None
    T5_QUESTION_AFTER_ALIGNER = False
    SEED_IN_NAME = False
    HEIGHT = WIDTH = 512              # reference :258-259

    def inputs(self):
        run = self.cfg.run_cfg
        return (run.get("question", None) if run.get("question", None) is not None else self.QUESTION,
                list(run.get("image_paths", None) or self.IMAGE_PATHS), list(run.get("texts", None) or self.TEXTS),
                list(run.get("image_names", None) or self.IMAGE_NAMES))

    def output_name(self, image_name, image_i, ckpt_id, seed):
        tail = f"_seed_{seed}" if self.SEED_IN_NAME else ""
        return f"{image_name}_output_embed_edit_4_flux_{image_i}_{ckpt_id}{tail}.png"

    def condition(self, language_model_inputs_i, question):
        """-> (prompt_embeds [1,T,4096], pooled [1,768]); reference ..._multi_image.py:240-248, ..._multi_image_input.py:305-319."""
        pe = language_model_inputs_i.unsqueeze(0).to(torch.bfloat16)
        if not self.T5_QUESTION_AFTER_ALIGNER:
            return pe, self.pooled_empty_prompt().to(torch.bfloat16)
        # [aligner tokens || T5(question)], and the pooled vector of the SECOND encode_prompt call: CLIP(question)
        if self.text is not None:
            t5, pooled = self.text.t5(question, 128, self.device), self.text.clip_pooled(question, self.device)
        else:
            t5, pooled, _ = self.pipe.encode_prompt(prompt=question, prompt_2=None, max_sequence_length=128)
        return torch.cat([pe, t5.to(torch.bfloat16)], dim=1), pooled.to(torch.bfloat16)

    def run(self):
        from thinkdiff.models.qwen2_vl import process_vision_info
        run = self.cfg.run_cfg
        out_dir = run["output_dir"]
        os.makedirs(out_dir, exist_ok=True)
        seed = run.seed + get_rank()
        if run.get("shard_prompts", False):
            # SURVEY.md 8(e) form of this driver: its work list is ONE request; it is planned on rank 0, broadcast, and taken by
            # rank 0 (jobs[rank::world]); the other ranks render nothing instead of repeating it under seed + rank
            work = dp.shard(dp.broadcast_work_list([(0, run.seed)] if get_rank() == 0 else None))
            if not work:
                dp.gather_results([])
                return []
            seed = work[0][1]
        ckpt_id = os.path.basename(self.cfg.model_cfg["ckpt"] or "")
        question, image_paths, texts, image_names = self.inputs()
        messages = build_messages(question, image_paths, texts, self.QUESTION_IN_CHAT, self.MAX_PIXELS)
        prompt = self.model.module.mllama_processor.apply_chat_template(messages, tokenize=False, add_generation_prompt=True, add_vision_id=True)
        print(prompt)
        image_data, _ = process_vision_info(messages)
        sample = {"prompt": prompt, "multi_modal_data": {"image": image_data}}
        embedding_type = self.cfg.model_cfg.get("embedding_type", "output_embed")
        with torch.no_grad():
            language_model_inputs, generated = self.model.module.get_embed(sample, embedding_type=embedding_type, max_new_tokens=128, need_process=False)
        for i, text in enumerate(generated):
            print(language_model_inputs[i].shape)
            print(text)
        written = []
        for img_i in range(1):                                    # reference :236
            with torch.no_grad():
                pe, pooled = self.condition(language_model_inputs[img_i], question)
            setup_seeds(seed)                                      # reference :252: the latents do not depend on how many tokens were sampled
            images = self.pipe(prompt_embeds=pe, pooled_prompt_embeds=pooled, num_images_per_prompt=1, height=self.HEIGHT, width=self.WIDTH,
                               num_inference_steps=28, guidance_scale=3.5).images
            for image_i, image in enumerate(images):
                path = f"{out_dir}/{self.output_name(image_names[img_i], image_i, ckpt_id, seed)}"
                image.save(path, format="PNG", compress_level=1)
                print(f"Saved image to {path}")
                written.append(path)
        if run.get("shard_prompts", False):
            every = dp.gather_results(written)
            return every if every is not None else written
        return written


def main(argv=None, driver_cls=LvlmMultiImageFluxDriver):
    args = parse_args(argv)
    cfg = Config(args)
    init_distributed_mode(cfg.run_cfg)
    setup_seeds(cfg.run_cfg.seed + get_rank())
    cfg.pretty_print()
    return driver_cls(cfg).run()


if __name__ == "__main__":
    main()
