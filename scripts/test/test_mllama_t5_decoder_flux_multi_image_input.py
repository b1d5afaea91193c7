"""ThinkDiff-LVLM image(s) as input + text instruction for FLUX -> driver on the MI355X path.

Mirror of the reference driver scripts/test/test_mllama_t5_decoder_flux_multi_image_input.py:80-340.  Differences from the
multi-image driver beside it, all kept:
  * the chat request holds only the (text_i, image_i) parts, each image capped at `max_pixels=65536` (:253-277) -- the
    question is NOT shown to the LVLM;
  * FLUX is conditioned on [aligner tokens || T5(question, 128 tokens)] (:305-319) and on the pooled vector of that second
    `encode_prompt` call, i.e. CLIP(question);
  * the output name carries the seed: `{image_name}_output_embed_edit_4_flux_{i}_{ckpt_id}_seed_{seed}.png` (:337-338).
Defaults = the reference's hard-coded inputs (:139, :236-246); override with run.question / image_paths / texts / image_names.
"""
from scripts.test.test_mllama_t5_decoder_flux_multi_image import LvlmMultiImageFluxDriver, main as _main


class LvlmMultiImageInputFluxDriver(LvlmMultiImageFluxDriver):
    QUESTION = "Reconstruct the texts in this image."
    IMAGE_PATHS = ["/root/dataset/minigpt4/MARIOEval/MARIOEval/LAIONEval4000/images/0.jpg"]
    TEXTS = [""]
    IMAGE_NAMES = ["LAIONEval4000_0"]
    QUESTION_IN_CHAT = False
    MAX_PIXELS = 65536
    T5_QUESTION_AFTER_ALIGNER = True
    SEED_IN_NAME = True


def main(argv=None):
    return _main(argv, driver_cls=LvlmMultiImageInputFluxDriver)


if __name__ == "__main__":
    main()
