"""ThinkDiff-CLIP two-image composition -> FLUX driver on the MI355X path (BASELINE config 5's launcher).

Mirror of the reference driver scripts/test/test_blip_vision_t5_decoder_flux.py:156-234, run by
runs/test_thinkdiff_clip_two_images.sh with configs/test_thinkdiff_clip_two_images.yaml.  It differs from the image+text
driver (scripts/test/test_blip_vision_t5_decoder_flux_text.py) in three places, all kept:
  * prompts come from `run.questions` x `run.questions_names` only -- there is no `prompt_json` mode (:145-146);
  * `run.use_image_name_as_output_name` is honoured in that mode: `{image_name}.png`, else
    `{image_name}_clip_t5_flux_{name}_seed_{seed}.png` (:161-164);
  * the PNG is written with PIL's default compression (`images[0].save(output_path)`, :233).
Token order per job: [aligner(img1), aligner(img2), ..., T5(prompt)] (:176-200); a list entry of `run.img_urls` is one
composition, its name the file stems joined with "_" (:133-140).

    python -m scripts.test.test_blip_vision_t5_decoder_flux --cfg-path configs/test_thinkdiff_clip_two_images.yaml \
        [--options run.synthetic=true run.flux_precision=fp8 run.shard_prompts=true ...]
"""
from scripts.test.test_blip_vision_t5_decoder_flux_text import ClipFluxDriver, main as _main


class ClipTwoImagesFluxDriver(ClipFluxDriver):
    TWO_IMAGE_DRIVER = True
    PNG_SAVE_KW = dict()            # reference :233 saves with the default compress_level


def main(argv=None):
    return _main(argv, driver_cls=ClipTwoImagesFluxDriver)


if __name__ == "__main__":
    main()
