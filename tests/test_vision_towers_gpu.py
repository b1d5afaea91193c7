"""GPU parity of the HIP vision towers against the modules the reference runs: transformers Blip2VisionModel
(blip_vision_t5_decoder.py:611-618) and Qwen2VisionTransformerPretrainedModel (inside vLLM in the reference).
Tiny random configs with the real head widths (88 and 80), bf16 on CPU.  Tolerance: relative RMSE <= 2e-2
(same bf16 rounding points, different fp32 summation order)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt())


def test_blip2_vision_matches_transformers(hip):
    from transformers import Blip2VisionConfig, Blip2VisionModel
    from thinkdiff.models.vision_towers import HipBlip2VisionModel
    torch.manual_seed(0)
    cfg = Blip2VisionConfig(hidden_size=704, intermediate_size=1408, num_hidden_layers=3, num_attention_heads=8, image_size=112,
                            patch_size=14, qkv_bias=True)            # 8 heads x 88 like EVA-ViT-g; 8x8 patches + CLS = 65 tokens
    ref = Blip2VisionModel(cfg).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "bias" in n:
                p.normal_(0, 0.05)
            elif p.dim() > 1 and "embedding" not in n:
                p.normal_(0, 0.04)       # config initializer_range is 1e-10: re-draw so the layers matter
            elif "norm" in n:
                p.normal_(1.0, 0.1)
    ref = ref.bfloat16()
    pix = torch.randn(2, 3, 112, 112)
    with torch.no_grad():
        want = ref(pixel_values=pix.bfloat16())
    tower = HipBlip2VisionModel(ref.state_dict(), num_heads=8, eps=cfg.layer_norm_eps)
    got = tower(pix)
    torch.cuda.synchronize()
    with torch.no_grad():
        skip_layers = ref.post_layernorm(ref.embeddings(pix.bfloat16()))
    assert _rel(skip_layers, want.last_hidden_state) > 0.3       # the encoder layers are not a no-op
    e1, e2 = _rel(got[0], want.last_hidden_state), _rel(got.pooler_output, want.pooler_output)
    print(f"BLIP-2 ViT rel-RMSE hidden {e1:.4f} pooled {e2:.4f}")
    assert got[0].shape == (2, 65, 704) and e1 < 2e-2 and e2 < 2e-2


def test_qwen2_vision_matches_transformers(hip):
    from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLVisionConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VisionTransformerPretrainedModel
    from thinkdiff.models.vision_towers import HipQwen2VisionTransformer
    torch.manual_seed(1)
    cfg = Qwen2VLVisionConfig(depth=3, embed_dim=320, hidden_size=256, num_heads=4, mlp_ratio=2, patch_size=14, temporal_patch_size=2,
                              spatial_merge_size=2, in_channels=3)    # 4 heads x 80 like the released tower
    ref = Qwen2VisionTransformerPretrainedModel(cfg).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "bias" in n:
                p.normal_(0, 0.05)
            elif p.dim() > 1:
                p.mul_(2.0)
    ref = ref.bfloat16()
    grid = torch.tensor([[1, 8, 12], [1, 6, 6]])                     # two images: 96 + 36 patches
    S = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
    patches = torch.randn(S, 3 * 2 * 14 * 14)
    with torch.no_grad():
        want = ref(patches.bfloat16(), grid_thw=grid)
    tower = HipQwen2VisionTransformer(ref.state_dict(), num_heads=4)
    got = tower(patches, grid)
    torch.cuda.synchronize()
    e1, e2 = _rel(got[0], want.last_hidden_state), _rel(got.pooler_output, want.pooler_output)
    print(f"Qwen2-VL ViT rel-RMSE hidden {e1:.4f} merged {e2:.4f}")
    assert got.pooler_output.shape == (S // 4, 256) and e1 < 2e-2 and e2 < 2e-2


def test_qwen2_vision_full_size_matches_transformers(hip):
    """The released Qwen2-VL-7B tower's SHAPE (32 layers, width 1280, 16 heads of 80, MLP x4, merger -> 3584; 675 M parameters, transformers' own
    random init) on config 3's image grid (224 x 224 -> 16 x 16 patches -> 64 merged tokens, the grid of tests/golden/full_depth_cfg3_lvlm7b.pt) plus
    a larger second image, against `Qwen2VisionTransformerPretrainedModel` itself run on the host in bf16 AND in fp32.  32 random layers amplify
    rounding noise, so the bar is the one the decoder's full-size test uses: HIP no further from the exact (fp32) result than 1.5 x the module's own
    bf16 run is."""
    from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLVisionConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VisionTransformerPretrainedModel
    from thinkdiff.models.vision_towers import HipQwen2VisionTransformer
    torch.manual_seed(3)
    cfg = Qwen2VLVisionConfig(depth=32, embed_dim=1280, hidden_size=3584, num_heads=16, mlp_ratio=4, patch_size=14, temporal_patch_size=2,
                              spatial_merge_size=2, in_channels=3)
    ref32 = Qwen2VisionTransformerPretrainedModel(cfg).eval()
    grid = torch.tensor([[1, 16, 16], [1, 20, 28]])                  # 256 + 560 patches -> 64 + 140 merged tokens
    S = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
    patches = torch.randn(S, 3 * 2 * 14 * 14).bfloat16()
    sd16 = {k: v.bfloat16() for k, v in ref32.state_dict().items()}
    with torch.no_grad():
        ref32.load_state_dict({k: v.float() for k, v in sd16.items()})       # the same bf16-representable weights on every side
        exact = ref32(patches.float(), grid_thw=grid).pooler_output
        want16 = ref32.bfloat16()(patches, grid_thw=grid).pooler_output
    tower = HipQwen2VisionTransformer(sd16, num_heads=16)
    got = tower(patches.float(), grid).pooler_output
    torch.cuda.synchronize()
    e_hip, e_ref, e_pair = _rel(got, exact), _rel(want16, exact), _rel(got, want16)
    print(f"Qwen2-VL-7B-shaped ViT, 204 merged tokens: HIP vs exact {e_hip:.4f}, transformers bf16 vs exact {e_ref:.4f}, HIP vs transformers bf16 {e_pair:.4f}")
    assert got.shape == (S // 4, 3584) and torch.isfinite(got.float()).all()
    assert e_hip < 1.5 * e_ref + 2e-3 and e_pair < 2.5 * e_ref + 2e-3


def test_rope_half_and_patchify_exact(hip):
    """Bit-level checks of the two data-movement kernels against torch on the same device."""
    from thinkdiff import _hip
    torch.manual_seed(2)
    S, H, hd = 37, 6, 80
    x = torch.randn(S, H * 128 + 64, device="cuda").bfloat16()
    ang = torch.rand(S, hd // 2, device="cuda") * 6.0
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    xv = x[:, :H * 128].reshape(S, H, 128).float()
    a, b = xv[..., :hd // 2], xv[..., hd // 2:hd]
    want = xv.clone()
    want[..., :hd // 2] = a * cos[:, None] - b * sin[:, None]
    want[..., hd // 2:hd] = b * cos[:, None] + a * sin[:, None]
    y = x.clone()
    _hip.rope_half(y, H, hd, cos, sin)
    torch.cuda.synchronize()
    assert torch.equal(y[:, H * 128:], x[:, H * 128:])
    got = y[:, :H * 128].reshape(S, H, 128).float()
    assert (got - want.bfloat16().float()).abs().max() <= 2 ** -6 * want.abs().max()     # <= 1 bf16 ulp (fma contraction)
    assert torch.equal(got[..., hd:], xv[..., hd:])
    from thinkdiff.models.vision_towers import vision_position_ids
    pos = vision_position_ids([[1, 8, 12], [2, 6, 6]])
    c2, s2 = _hip.vision_rope_table(pos.to("cuda", torch.int32).contiguous(), hd)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float32) / (hd // 2)))
    ang2 = (pos.float()[:, :, None] * inv).flatten(1)
    assert (c2.cpu() - ang2.cos()).abs().max() < 2e-5 and (s2.cpu() - ang2.sin()).abs().max() < 2e-5
    pix = torch.randn(3, 28, 42, device="cuda")
    out = _hip.patchify(pix, 14, 640)
    torch.cuda.synchronize()
    ref = torch.nn.functional.unfold(pix[None], 14, stride=14)[0].T.bfloat16()          # [6, 588], column order c, iy, ix
    assert torch.equal(out[:, :588], ref) and not out[:, 588:].any()
    src = torch.randn(5, 1176, device="cuda")
    cp = _hip.cast_pad_rows(src, 1216)
    torch.cuda.synchronize()
    assert torch.equal(cp[:, :1176], src.bfloat16()) and not cp[:, 1176:].any()


def test_image_request_hidden_states_match_qwen2vl_model(hip):
    """An image + text request end to end (processor -> ViT -> placeholder splice -> M-RoPE -> decoder) against
    transformers Qwen2VLModel on the same pixels: the `model.norm` hidden states the reference feeds the aligner."""
    import numpy as np
    from PIL import Image
    from transformers import Qwen2VLImageProcessor
    from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLModel
    from thinkdiff.models.mllama_vllm_t5_embed_decoder_2 import MllamaVllmT5EmbedDecoderForConditionalGeneration_5
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig
    from thinkdiff.models.vision_towers import HipQwen2VisionTransformer
    IMG, VS, VE = 1000, 1002, 1003
    cfg = Qwen2VLConfig(
        text_config=dict(hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512,
                         vocab_size=1024, max_position_embeddings=4096, rms_norm_eps=1e-6,
                         rope_parameters={"rope_type": "default", "mrope_section": [16, 24, 24], "rope_theta": 1e6}),
        vision_config=dict(depth=2, embed_dim=320, hidden_size=512, num_heads=4, mlp_ratio=2),
        image_token_id=IMG, video_token_id=1001, vision_start_token_id=VS, vision_end_token_id=VE)
    torch.manual_seed(3)
    ref = Qwen2VLModel(cfg).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if p.dim() > 1 and "embed_tokens" not in n:
                p.mul_(2.0)
    ref = ref.bfloat16()
    sd = ref.state_dict()
    tsd = {"model." + k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    tsd["lm_head.weight"] = tsd["model.embed_tokens.weight"]
    m = MllamaVllmT5EmbedDecoderForConditionalGeneration_5(
        Qwen2VLTextConfig(hidden_size=512, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2, intermediate_size=512,
                          vocab_size=1024, tie_word_embeddings=True),
        vllm_config={"max_model_len": 512}, hidden_size=4096,
        visual=HipQwen2VisionTransformer({k: v for k, v in sd.items() if k.startswith("visual.")}, num_heads=4),
        image_processor=Qwen2VLImageProcessor(min_pixels=56 * 56, max_pixels=28 * 28 * 64), image_token_id=IMG)
    m.mllama.load_state_dict(tsd, strict=False)
    rng = np.random.default_rng(0)
    image = Image.fromarray((rng.random((200, 300, 3)) * 255).astype(np.uint8))
    prompt = [5, 6, 7, VS, IMG, VE, 8, 9, 10, 11]
    ids, emb, pos = m._splice_images(prompt, image)
    hid, _ = m.mllama.forward(pos, None, emb)
    torch.cuda.synchronize()
    feats = m.image_processor(images=[image], return_tensors="pt")
    inp = torch.tensor([ids])
    with torch.no_grad():
        want = ref(input_ids=inp, pixel_values=feats["pixel_values"], image_grid_thw=feats["image_grid_thw"],
                   mm_token_type_ids=(inp == IMG).int()).last_hidden_state[0]
    n_img = int(feats["image_grid_thw"].prod()) // 4
    assert len(ids) == len(prompt) - 1 + n_img and hid.shape == want.shape
    e = _rel(hid, want)
    print(f"image request: {len(ids)} tokens ({n_img} vision) rel-RMSE {e:.4f}")
    assert e < 2e-2
    # and through get_embed (aligner output shape, forced continuation)
    embeds, texts = m.get_embed([{"prompt_token_ids": prompt, "multi_modal_data": {"image": image}}], embedding_type="both",
                                need_process=False, forced_output_ids=[[3, 4, 5]])
    assert embeds[0].shape == (len(ids) + 3, 4096) and texts == ["3 4 5"]


def test_qwen2vl_checkpoint_directory_loaders(hip, tmp_path):
    """A Hugging Face Qwen2-VL directory (config.json with vision_config, sharded safetensors holding `model.*`, `visual.*`,
    `lm_head.*`) -> Qwen2VLTextEngine.load_pretrained + HipQwen2VisionTransformer.from_pretrained, both the older flat names
    and the newer `model.language_model.* / model.visual.*` nesting; results equal the state-dict constructors."""
    import json
    from safetensors.torch import save_file
    from oracle import qwen2vl_ref as Q
    from thinkdiff.models.qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine
    from thinkdiff.models.vision_towers import HipQwen2VisionTransformer
    from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLVisionConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VisionTransformerPretrainedModel
    torch.manual_seed(4)
    cfg = Q.tiny_config()
    sd = Q.init_weights(cfg, seed=3)
    vcfg = Qwen2VLVisionConfig(depth=2, embed_dim=320, hidden_size=cfg.hidden, num_heads=4, mlp_ratio=2)
    vsd = {k: v.bfloat16() for k, v in Qwen2VisionTransformerPretrainedModel(vcfg).state_dict().items()}
    tc = Qwen2VLTextConfig(hidden_size=cfg.hidden, num_hidden_layers=cfg.num_layers, num_attention_heads=cfg.num_heads,
                           num_key_value_heads=cfg.num_kv_heads, intermediate_size=cfg.intermediate, vocab_size=cfg.vocab)
    ref_e = Qwen2VLTextEngine(tc, max_model_len=128)
    ref_e.load_state_dict(sd)
    ref_v = HipQwen2VisionTransformer(vsd, num_heads=4)
    ids = torch.randint(0, cfg.vocab, (33,), dtype=torch.int32)
    patches, grid = torch.randn(48, 1176), [[1, 6, 8]]
    want_h = ref_e.forward(ref_e.text_position_ids(33), ids)[0].clone()
    want_v = ref_v(patches, grid).pooler_output.clone()
    for style in ("flat", "nested"):
        root = tmp_path / style
        root.mkdir()
        tp, vp = ("model.", "visual.") if style == "flat" else ("model.language_model.", "model.visual.")
        full = {(tp + k[len("model."):] if k.startswith("model.") else k): v.contiguous() for k, v in sd.items()}
        full.update({vp + k: v.contiguous() for k, v in vsd.items()})
        keys = sorted(full)
        save_file({k: full[k] for k in keys[::2]}, str(root / "model-00001-of-00002.safetensors"))
        save_file({k: full[k] for k in keys[1::2]}, str(root / "model-00002-of-00002.safetensors"))
        (root / "config.json").write_text(json.dumps({"model_type": "qwen2_vl", "vision_config": {"num_heads": 4, "spatial_merge_size": 2, "hidden_act": "quick_gelu"}}))
        e = Qwen2VLTextEngine(tc, max_model_len=128).load_pretrained(str(root))
        v = HipQwen2VisionTransformer.from_pretrained(str(root))
        got_h = e.forward(e.text_position_ids(33), ids)[0]
        got_v = v(patches, grid).pooler_output
        torch.cuda.synchronize()
        assert torch.equal(got_h, want_h) and torch.equal(got_v, want_v), style


def test_blip2_vision_on_disk_bias_names(hip, tmp_path):
    """BLIP-2 checkpoints store the attention bias as `q_bias` / `v_bias` (k has none) under `vision_model.`; the in-memory
    module fuses them into `qkv.bias`.  Both spellings, and HipBlip2VisionModel.from_pretrained on a directory, agree."""
    import json
    from safetensors.torch import save_file
    from transformers import Blip2VisionConfig, Blip2VisionModel
    from thinkdiff.models.vision_towers import HipBlip2VisionModel
    torch.manual_seed(7)
    cfg = Blip2VisionConfig(hidden_size=704, intermediate_size=1408, num_hidden_layers=2, num_attention_heads=8, image_size=112, patch_size=14, qkv_bias=True)
    ref = Blip2VisionModel(cfg).eval()
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if "qkv.bias" in n:
                p.normal_(0, 0.05)
                p[704:1408] = 0          # k carries no bias in the checkpoints
            elif p.dim() > 1 and "embedding" not in n:
                p.normal_(0, 0.04)
    sd = {k: v.bfloat16() for k, v in ref.state_dict().items()}
    disk = {}
    for k, v in sd.items():
        if k.endswith("self_attn.qkv.bias"):
            disk["vision_model." + k.replace("qkv.bias", "q_bias")] = v[:704].clone()
            disk["vision_model." + k.replace("qkv.bias", "v_bias")] = v[1408:].clone()
        else:
            disk["vision_model." + k] = v.contiguous()
    disk["language_projection.weight"] = torch.zeros(8, 8).bfloat16()      # other parts of a BLIP-2 checkpoint are ignored
    save_file(disk, str(tmp_path / "model.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps({"vision_config": {"num_attention_heads": 8, "patch_size": 14, "layer_norm_eps": cfg.layer_norm_eps}}))
    pix = torch.randn(1, 3, 112, 112)
    a = HipBlip2VisionModel(sd, num_heads=8, eps=cfg.layer_norm_eps)(pix)[0]
    b = HipBlip2VisionModel(disk, num_heads=8, eps=cfg.layer_norm_eps)(pix)[0]
    c = HipBlip2VisionModel.from_pretrained(str(tmp_path))(pix)[0]
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, c)


def test_qwen2_preprocessing_on_device_equals_the_image_processor(hip):
    """QwenChatFrontend._preprocess_on_device (host: RGB conversion + smart_resize / PIL resize; device: td_qwen2_patchify_u8 with
    the processor's own rescale / normalize table) against transformers' Qwen2VLImageProcessor on the same PIL images: grids
    equal, patch rows bit-equal after the bf16 cast the tower applies, zero padding columns; modes L / RGBA, up- and down-sizing."""
    import numpy as np
    from PIL import Image
    from transformers import Qwen2VLImageProcessor
    from thinkdiff.models.qwen2_vl import QwenChatFrontend
    from thinkdiff.models.vision_towers import HipQwen2VisionTransformer

    class Front(QwenChatFrontend):
        pass
    f = Front()
    f.visual = HipQwen2VisionTransformer.from_random(embed_dim=320, depth=1, num_heads=4, mlp_ratio=2, out_hidden=256, seed=1)
    f.image_processor = Qwen2VLImageProcessor(min_pixels=56 * 56, max_pixels=28 * 28 * 320)
    assert type(f.image_processor).__name__ == "Qwen2VLImageProcessorPil"      # torchvision is absent: the PIL pipeline, the one mirrored
    rng = np.random.default_rng(0)
    imgs = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for h, w in [(375, 500), (28, 30), (1200, 900), (224, 224)]]
    imgs.append(Image.fromarray(rng.integers(0, 256, (90, 130), dtype=np.uint8), mode="L"))
    imgs.append(Image.fromarray(rng.integers(0, 256, (64, 200, 4), dtype=np.uint8), mode="RGBA"))
    got = f._preprocess_on_device(imgs)
    assert got is not None
    want = f.image_processor(images=imgs, return_tensors="pt")
    assert got["image_grid_thw"] == want["image_grid_thw"].tolist()
    K = want["pixel_values"].shape[1]
    pv = got["pixel_values"].cpu()
    assert pv.shape == (want["pixel_values"].shape[0], f.visual.padded_patch_dim) and pv.dtype == torch.bfloat16
    assert torch.equal(pv[:, :K], want["pixel_values"].bfloat16()) and torch.count_nonzero(pv[:, K:]) == 0
    # and the tower gives the same tokens either way
    a = f.visual(got["pixel_values"], got["image_grid_thw"]).pooler_output
    b = f.visual(want["pixel_values"], want["image_grid_thw"]).pooler_output
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # a processor of any other kind is called as is
    f.image_processor = lambda images, return_tensors="pt": want
    assert f._preprocess_on_device(imgs) is None
