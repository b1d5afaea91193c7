"""Host-side logic that needs no GPU: config loader on the reference's YAML content, registry,
schedule scalars vs the oracle, known-answer values from SURVEY.md 8c(4)."""
import argparse
import os

import numpy as np
import pytest
import torch

from oracle import flux_ref as R

HERE = os.path.dirname(os.path.abspath(__file__))


def test_known_answer_scalars():
    assert abs(R.calculate_shift(4096) - 1.15) < 1e-12
    assert abs(R.calculate_shift(256) - 0.5) < 1e-12
    s = R.make_sigmas(28, 4096)
    assert s.dtype == np.float32 and len(s) == 29 and s[0] == 1.0 and s[-1] == 0.0
    # shifted sigma_1 = e^1.15 / (e^1.15 + (28/27 - 1))
    assert abs(float(s[1]) - np.exp(1.15) / (np.exp(1.15) + (28 / 27 - 1))) < 1e-6
    assert np.all(np.diff(s) < 0)


def test_schedule_matches_oracle_and_effective_scalars():
    from thinkdiff.models.flux_prompt import FlowMatchEulerSchedule
    from thinkdiff.models.flux_transformer import effective_scalar
    for n, seq in [(4, 256), (28, 4096), (28, 1024), (50, 4096)]:
        assert np.array_equal(FlowMatchEulerSchedule.sigmas(n, seq), R.make_sigmas(n, seq))
    for t in [1000.0, 967.3, 500.0, 3.7]:
        assert effective_scalar(t, torch.bfloat16) == R.effective_timestep(t, torch.bfloat16)
        assert abs(effective_scalar(t, torch.float32) - t) < 1e-3
    assert effective_scalar(3500.0, torch.bfloat16) == 3504.0  # bf16(3.5)*1000 rounds to the 16-grid


def test_pack_unpack_roundtrip_and_ids():
    x = torch.arange(2 * 16 * 8 * 12, dtype=torch.float32).reshape(2, 16, 8, 12)
    p = R.pack_latents(x)
    assert p.shape == (2, 24, 64)
    assert torch.equal(R.unpack_latents(p, 8, 12), x)
    # token (i,j), column c*4 + di*2 + dj holds x[c, 2i+di, 2j+dj]
    assert p[0, 1 * 6 + 2, 5 * 4 + 2 + 1] == x[0, 5, 2 * 1 + 1, 2 * 2 + 1]
    ids = R.latent_image_ids(3, 4)
    assert ids.shape == (12, 3) and ids[5].tolist() == [0.0, 1.0, 1.0]


def test_config_loads_driver_keys_fixture(tmp_path):
    import thinkdiff.models  # noqa: F401  registers the archs
    from thinkdiff import tasks
    from thinkdiff.common.config import Config
    y = tmp_path / "clip.yaml"
    y.write_text(open(os.path.join(HERE, "golden", "thinkdiff_clip_driver_keys.yaml")).read())
    cfg = Config(argparse.Namespace(cfg_path=str(y), options=["run.flux_height=256", "run.flux_num_inference_steps=4"]))
    run = cfg.run_cfg
    assert run.flux_height == 256 and run["flux_width"] == 1024 and run.flux_num_inference_steps == 4
    assert run.get("img_folder", None) is None and type(run.img_urls) == list
    assert run.seed == 42 and run.guidance_scale == 3.5 and run.flux_max_sequence_length == 128
    assert cfg.model_cfg.arch == "blip-vision-t5-decoder" and cfg.model_cfg.mm_projector_type == "mlp2x_gelu_t5_norm"
    assert "laion" in cfg.datasets_cfg  # unknown dataset kept, not a crash (reference config.py:99-104 would)
    assert tasks.setup_task(cfg) is not None


def test_registry_surface():
    from thinkdiff.common.registry import registry
    import thinkdiff.models  # noqa: F401
    import thinkdiff.runners  # noqa: F401
    assert registry.get_model_class("blip-vision-t5-decoder") is not None
    assert registry.get_model_class("nope") is None
    assert registry.get_runner_class("runner_clip_t5") is not None
    registry.register_path("p_test", "/a")
    with pytest.raises(KeyError):
        registry.register_path("p_test", "/b")


def test_flux_param_names_cover_diffusers_state_dict():
    """FLUX.1-dev must total 11.9 B parameters; the same names index the HIP engine's arena."""
    shapes = R.param_shapes(R.FluxConfig())
    total = sum(int(np.prod(s)) for s in shapes.values())
    assert abs(total / 1e9 - 11.90) < 0.01
    assert "transformer_blocks.18.attn.norm_added_k.weight" in shapes
    assert shapes["single_transformer_blocks.37.proj_out.weight"] == (3072, 15360)


@pytest.mark.skipif(not os.path.exists("/root/reference/configs"), reason="reference tree not mounted (GPU box)")
@pytest.mark.parametrize("name", ["test_thinkdiff_clip_image_text.yaml", "test_thinkdiff_clip_two_images.yaml"])
def test_config_loads_unchanged_reference_yaml(name):
    """Read-only study of the mounted reference: its committed YAMLs must load as they are."""
    import thinkdiff.models  # noqa: F401
    from thinkdiff.common.config import Config
    cfg = Config(argparse.Namespace(cfg_path=os.path.join("/root/reference/configs", name), options=None))
    assert cfg.run_cfg.flux_num_inference_steps == 28 and cfg.run_cfg.guidance_scale == 3.5
    assert cfg.model_cfg.arch == "blip-vision-t5-decoder"


def test_mrope_position_ids_match_transformers_get_rope_index():
    """Qwen2VLTextEngine.expand_image_placeholders / mrope_position_ids vs transformers Qwen2VLModel.get_rope_index
    (two images of different grids, text before / between / after)."""
    import torch
    from transformers.models.qwen2_vl.configuration_qwen2_vl import Qwen2VLConfig
    from transformers.models.qwen2_vl.modeling_qwen2_vl import Qwen2VLModel
    from thinkdiff.models.qwen2_vl import Qwen2VLTextEngine as E
    cfg = Qwen2VLConfig(text_config=dict(hidden_size=64, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=1,
                                         intermediate_size=64, vocab_size=1024),
                        vision_config=dict(depth=1, embed_dim=32, hidden_size=64, num_heads=2, mlp_ratio=1),
                        image_token_id=1000, video_token_id=1001, vision_start_token_id=1002, vision_end_token_id=1003)
    m = Qwen2VLModel(cfg)
    grid = [[1, 8, 12], [1, 6, 6]]
    ids = E.expand_image_placeholders([5, 6, 7, 1002, 1000, 1003, 8, 9, 1002, 1000, 1003, 10, 11, 12], grid, 2, 1000)
    assert len(ids) == 12 + 24 + 9 and E.expand_image_placeholders(ids, grid, 2, 1000) == ids
    pos = E.mrope_position_ids(ids, grid, 2, 1000)
    inp = torch.tensor([ids])
    want, _ = m.get_rope_index(inp, mm_token_type_ids=(inp == 1000).int(), image_grid_thw=torch.tensor(grid))
    assert torch.equal(want[:, 0].int(), pos)
    try:
        E.expand_image_placeholders([1000], grid, 2, 1000)
    except ValueError:
        pass
    else:
        raise AssertionError("placeholder / image count mismatch must raise")
