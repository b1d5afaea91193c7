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


def test_driver_groups_images_in_flight_without_changing_them(hip, tmp_path):
    """4 (image, prompt) jobs incl. a two-image composition (BASELINE config 5's driver shape): rendering in groups of 3
    (engine contexts on separate streams) writes the same PNG bytes as one image at a time, in bf16 and in fp8."""
    import hashlib
    from scripts.test import test_blip_vision_t5_decoder_flux_text as drv
    imgs = []
    for k, col in enumerate([(120, 80, 40), (10, 200, 90), (250, 250, 0)]):
        p = tmp_path / f"img{k}.jpg"
        Image.new("RGB", (64 + 8 * k, 64), col).save(p)
        imgs.append(str(p))
    base = ["--cfg-path", os.path.join(HERE, "golden", "thinkdiff_clip_driver_keys.yaml"), "--options",
            "run.synthetic=true", "run.synthetic_tiny=true", "run.flux_height=128", "run.flux_width=128",
            "run.flux_num_inference_steps=2", f"run.img_urls=[{imgs[0]},{imgs[1]},{imgs[2]},[{imgs[0]},{imgs[1]}]]",
            "run.questions=[a red apple]", "run.questions_names=[apple]", "run.prompt_json=", "run.use_image_name_and_prompt_as_output_name=false", "model.ckpt="]
    digests = {}
    for prec in ("bf16", "fp8"):
        for G in (1, 3):
            out = tmp_path / f"out_{prec}_{G}"
            written = drv.main(base + [f"run.output_dir={out}", f"run.images_in_flight={G}", f"run.flux_precision={prec}"])
            assert len(written) == 4 and os.path.basename(written[3]) == "img0_img1_clip_t5_flux_apple_seed_42.png"   # reference ..._flux_text.py:254
            digests[prec, G] = [hashlib.sha256(open(w, "rb").read()).hexdigest() for w in written]
        assert digests[prec, 1] == digests[prec, 3]
    assert digests["bf16", 1] != digests["fp8", 1]
    # the whole 8-bit path through the config surface: int8 Linears under history scales + the e4m3 joint attention
    out = tmp_path / "out_int8_attn8"
    written = drv.main(base + [f"run.output_dir={out}", "run.images_in_flight=2", "run.flux_precision=int8", "run.flux_act_scales=history", "run.flux_smoothing=true", "run.flux_attention=fp8"])
    assert len(written) == 4 and all(os.path.getsize(w) > 0 for w in written)
    assert [hashlib.sha256(open(w, "rb").read()).hexdigest() for w in written] != digests["bf16", 1]


def test_lvlm_image_instruction_driver_writes_png(hip, tmp_path):
    """BASELINE config 3 in miniature: image + instruction -> Qwen2-VL ViT + decoder (sampled tokens, hidden states) ->
    aligner -> FLUX -> VAE -> PNG, through the reference's config surface and output naming."""
    from scripts.test import test_mllama_t5_decoder_flux as drv
    img = tmp_path / "dot_image.jpeg"
    Image.new("RGB", (280, 196), (30, 90, 200)).save(img)
    out = tmp_path / "out"
    argv = ["--cfg-path", os.path.join(HERE, "golden", "thinkdiff_lvlm_driver_keys.yaml"), "--options",
            "run.synthetic=true", "run.synthetic_tiny=true", "run.distributed=false", f"run.img_urls=[{img}]", f"run.output_dir={out}",
            "run.flux_height=256", "run.flux_width=256", "run.flux_num_inference_steps=2", "model.ckpt=",
            "model.vllm_config.max_model_len=1024", "model.vllm_config.max_tokens=16", "model.vllm_config.min_tokens=16",
            "model.text_config={hidden_size: 512, num_hidden_layers: 2, num_attention_heads: 4, num_key_value_heads: 2, intermediate_size: 1024, vocab_size: 152064}"]
    written = drv.main(argv)
    assert [os.path.basename(w) for w in written] == ["dot_image_output_embed_flux_0.png"]
    im = Image.open(written[0])
    assert im.size == (256, 256) and im.mode == "RGB"


def test_precompute_job_end_to_end(hip, tmp_path):
    """BASELINE config 4 in miniature: scripts/generate_embedding_webdataset over two input shards -> output shards with
    jpg, json (+ generated text / token ids) and the model.norm input / output hidden states as torch.save bytes."""
    import io
    import sys
    import torch
    sys.path.insert(0, HERE)
    from test_precompute_cpu import _make_input_shards
    from scripts import generate_embedding_webdataset as job
    from thinkdiff.datasets import wds_io
    idx, n = _make_input_shards(str(tmp_path), n_shards=2, per_shard=3)
    out = tmp_path / "emb"
    argv = ["--cfg-path", os.path.join(HERE, "golden", "qwen2_vl_embed_keys.yaml"), "--options", "run.synthetic=true", "run.synthetic_tiny=true",
            f"datasets.cc_sbu_mllama_vllm_process_wids.build_info.storage={idx}", "datasets.cc_sbu_mllama_vllm_process_wids.batch_size=4",
            f"run.output_shard_path=[{out},'%06d.tar',7]", "model.vllm_config.max_model_len=1024", "model.vllm_config.max_tokens=12",
            "model.vllm_config.min_tokens=12", "model.vllm_config.ignore_eos=true",
            "model.text_config={hidden_size: 512, num_hidden_layers: 2, num_attention_heads: 4, num_key_value_heads: 2, intermediate_size: 1024, vocab_size: 152064}"]
    res = job.main(argv)
    stats = res[0] if isinstance(res, list) else res
    assert stats["samples"] == n and os.path.basename(stats["shards"][0]["url"]) == "000007.tar"
    seen = {s["__key__"]: s for sh in stats["shards"] for s in wds_io.read_tar_samples(sh["url"])}
    assert sorted(seen) == [f"sample{k:06d}" for k in range(n)]
    one = seen["sample000002"]
    js = one[".json"]
    assert len(js["output_token_ids"]) == 12 and js["input_prompt"].startswith("<|im_start|>system") and "<|image_pad|>" in js["input_prompt"]
    oe = torch.load(io.BytesIO(one[".model.norm.output_embed.pth"])) if isinstance(one[".model.norm.output_embed.pth"], bytes) else one[".model.norm.output_embed.pth"]
    ie = torch.load(io.BytesIO(one[".model.norm.input_embed.pth"])) if isinstance(one[".model.norm.input_embed.pth"], bytes) else one[".model.norm.input_embed.pth"]
    assert oe.shape == (12, 512) and oe.dtype == torch.bfloat16 and ie.shape == (len(js["input_prompt_token_ids"]), 512)
    assert js["input_prompt_token_ids"].count(151655) > 1            # the image placeholder was expanded to the merged vision tokens


def test_two_image_driver_entry_point(hip, tmp_path):
    """BASELINE config 5's launcher (reference scripts/test/test_blip_vision_t5_decoder_flux.py): questions mode honours
    use_image_name_as_output_name (:161-162); without it the name is {image}_clip_t5_flux_{name}_seed_{seed}.png (:164)."""
    from scripts.test import test_blip_vision_t5_decoder_flux as drv
    imgs = []
    for k, col in enumerate([(200, 30, 40), (10, 20, 190)]):
        p = tmp_path / f"pic{k}.jpg"
        Image.new("RGB", (80, 64 + 8 * k), col).save(p)
        imgs.append(str(p))
    base = ["--cfg-path", os.path.join(HERE, "golden", "thinkdiff_clip_driver_keys.yaml"), "--options",
            "run.synthetic=true", "run.synthetic_tiny=true", "run.flux_height=128", "run.flux_width=128", "run.flux_num_inference_steps=2",
            f"run.img_urls=[[{imgs[0]},{imgs[1]}]]", "run.questions=['']", "run.questions_names=['null']", "run.prompt_json=",
            "run.use_image_name_and_prompt_as_output_name=false", "model.ckpt="]
    out = tmp_path / "o1"
    written = drv.main(base + [f"run.output_dir={out}", "run.use_image_name_as_output_name=true"])
    assert [os.path.basename(w) for w in written] == ["pic0_pic1.png"] and Image.open(written[0]).size == (128, 128)
    out = tmp_path / "o2"
    written = drv.main(base + [f"run.output_dir={out}"])
    assert [os.path.basename(w) for w in written] == ["pic0_pic1_clip_t5_flux_null_seed_42.png"]
    # sharded mode on one rank: same job, its own seed (seed + job index = 42) in the name
    out = tmp_path / "o3"
    written = drv.main(base + [f"run.output_dir={out}", "run.shard_prompts=true"])
    assert [os.path.basename(w) for w in written] == ["pic0_pic1_clip_t5_flux_null_seed_42.png"]


def test_lvlm_multi_image_drivers(hip, tmp_path):
    """Reference scripts/test/test_mllama_t5_decoder_flux_multi_image.py and ..._multi_image_input.py in miniature: chat
    request with two pictures (add_vision_id) -> get_embed(need_process=False) -> aligner -> FLUX 512^2; the input variant
    conditions on [aligner tokens || T5(question)] and CLIP(question)."""
    import torch
    from scripts.test import test_mllama_t5_decoder_flux_multi_image as mi
    from scripts.test import test_mllama_t5_decoder_flux_multi_image_input as mii
    from thinkdiff.common.config import Config
    imgs = []
    for k, col in enumerate([(250, 250, 250), (20, 30, 220)]):
        p = tmp_path / f"car{k}.jpg"
        Image.new("RGB", (140, 112), col).save(p)
        imgs.append(str(p))
    common = ["--cfg-path", os.path.join(HERE, "golden", "thinkdiff_lvlm_driver_keys.yaml"), "--options",
              "run.synthetic=true", "run.synthetic_tiny=true", "run.distributed=false", "model.ckpt=/ckpts/thinkdiff_lvlm.pth",
              "model.vllm_config.max_model_len=1024", "model.vllm_config.max_tokens=16", "model.vllm_config.min_tokens=16",
              "model.text_config={hidden_size: 512, num_hidden_layers: 2, num_attention_heads: 4, num_key_value_heads: 2, intermediate_size: 1024, vocab_size: 152064}"]
    out = tmp_path / "mi"
    written = mi.main(common + [f"run.output_dir={out}", f"run.image_paths=[{imgs[0]},{imgs[1]}]"])
    assert [os.path.basename(w) for w in written] == ["car_white_blue_red_output_embed_edit_4_flux_0_thinkdiff_lvlm.pth.png"]   # reference :267
    assert Image.open(written[0]).size == (512, 512)
    # same seed, same request -> the re-seed before the FLUX call (:252) makes a second run reproduce the image bit for bit
    out2 = tmp_path / "mi2"
    again = mi.main(common + [f"run.output_dir={out2}", f"run.image_paths=[{imgs[0]},{imgs[1]}]"])
    assert open(written[0], "rb").read() == open(again[0], "rb").read()

    out = tmp_path / "mii"
    argv = common + [f"run.output_dir={out}", f"run.image_paths=[{imgs[0]}]", "run.image_names=[LAIONEval4000_0]"]
    d = mii.LvlmMultiImageInputFluxDriver(Config(mi.parse_args(argv)))
    lm = torch.randn(16, 4096, device="cuda").bfloat16()
    pe, pooled = d.condition(lm, d.QUESTION)
    assert pe.shape == (1, 16 + 128, 4096) and pooled.shape == (1, 768)
    assert torch.equal(pe[0, :16], lm) and torch.equal(pe[0, 16:], d.text.t5(d.QUESTION, 128, d.device)[0].to(torch.bfloat16))   # aligner first
    assert torch.equal(pooled, d.text.clip_pooled(d.QUESTION, d.device).to(torch.bfloat16))                                     # CLIP(question)
    written = d.run()
    assert [os.path.basename(w) for w in written] == ["LAIONEval4000_0_output_embed_edit_4_flux_0_thinkdiff_lvlm.pth_seed_42.png"]   # reference :337


def test_lvlm_embed_export_drivers(hip, tmp_path):
    """Reference scripts/test/test_mllama_t5_decoder_flux_embed.py, ..._embed_multi_image.py, ..._embed_multi_image_batch.py
    and ..._multi_image_input_embed.py in miniature: the aligner output lands on disk as {name}.pth + {name}.json (or, for the
    text-only driver, conditions FLUX); the batched export writes the same tensors as the one-at-a-time export."""
    import io
    import json

    import torch
    from scripts.test import test_mllama_t5_decoder_flux_embed as emb
    from scripts.test import test_mllama_t5_decoder_flux_embed_multi_image as mi
    from scripts.test import test_mllama_t5_decoder_flux_embed_multi_image_batch as mib
    from scripts.test import test_mllama_t5_decoder_flux_multi_image_input_embed as txt
    common = ["--cfg-path", os.path.join(HERE, "golden", "thinkdiff_lvlm_driver_keys.yaml"), "--options",
              "run.synthetic=true", "run.synthetic_tiny=true", "run.distributed=false", "model.ckpt=/ckpts/thinkdiff_lvlm.pth",
              "model.vllm_config.max_model_len=1024", "model.vllm_config.max_tokens=16", "model.vllm_config.min_tokens=16",
              "model.text_config={hidden_size: 512, num_hidden_layers: 2, num_attention_heads: 4, num_key_value_heads: 2, intermediate_size: 1024, vocab_size: 152064}"]
    src = tmp_path / "in"
    src.mkdir()
    imgs = []
    for k, col in enumerate([(250, 250, 250), (20, 30, 220), (200, 30, 20)]):
        p = src / f"car{k}.jpg"
        Image.new("RGB", (140, 112), col).save(p)
        (src / f"car{k}.json").write_text(json.dumps({"caption": f"car {k}"}))
        imgs.append(str(p))

    # one image + run.prompt -> car{k}.pth / car{k}.json; a rendered car1.png in the output directory masks car1 (:146-150)
    out = tmp_path / "emb"
    out.mkdir()
    Image.new("RGB", (8, 8)).save(out / "car1.png")
    written = emb.main(common + [f"run.output_dir={out}", f"run.image_folder={src}", "run.prompt=Describe the car."])
    assert sorted(os.path.basename(w) for w in written) == ["car0.json", "car0.pth", "car2.json", "car2.pth"]
    e0 = torch.load(io.BytesIO(open(out / "car0.pth", "rb").read()))
    assert e0.device.type == "cpu" and e0.shape == (16, 4096) and torch.isfinite(e0.float()).all()
    j0 = json.load(open(out / "car0.json"))
    assert list(j0) == ["caption", "generated_text", "prompt"] and j0["prompt"] == "Describe the car." and j0["caption"] == "car 0"

    # interleaved word / picture tasks, one at a time and in batches of 2
    tasks_dir = tmp_path / "tasks"
    tasks_dir.mkdir()
    for k in range(3):
        (tasks_dir / f"task{k}.json").write_text(json.dumps({"text_inputs": ["white: ", "blue: ", "red: "][k:] + ["green: "], "image_inputs": imgs[k:][:2]}))
    one, many = tmp_path / "one", tmp_path / "many"
    opts = [f"run.image_folder={tasks_dir}", "run.prompt=What is the next picture?", "run.max_pixels=65536"]
    w1 = mi.main(common + [f"run.output_dir={one}"] + opts)
    assert sorted(os.path.basename(w) for w in w1) == sorted(f"task{k}.{x}" for k in range(3) for x in ("pth", "json"))
    assert mi.main(common + [f"run.output_dir={one}"] + opts) == []                       # every .pth exists -> nothing to do
    wb = mib.main(common + [f"run.output_dir={many}", "run.batch_size=2"] + opts)
    assert sorted(os.path.basename(w) for w in wb) == sorted(os.path.basename(w) for w in w1)
    for k in range(3):
        a, b = torch.load(one / f"task{k}.pth"), torch.load(many / f"task{k}.pth")
        assert a.shape == b.shape == (16, 4096)
        ja, jb = json.load(open(one / f"task{k}.json")), json.load(open(many / f"task{k}.json"))
        assert ja["prompt"] == jb["prompt"] == "What is the next picture?" and ja["text_inputs"] == jb["text_inputs"]

    # text-only prompt -> aligner tokens padded to run.max_tokens -> FLUX 512^2
    out = tmp_path / "txt"
    written = txt.main(common + [f"run.output_dir={out}", "run.max_tokens=24"])
    assert [os.path.basename(w) for w in written] == ["skateboard_edit_4_flux_output_embed_0_thinkdiff_lvlm.pth.png"]   # reference :289
    assert Image.open(written[0]).size == (512, 512)
