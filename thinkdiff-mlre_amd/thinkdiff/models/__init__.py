"""Model registry (reference thinkdiff/models/__init__.py): importing this package registers the archs."""
from ..common.registry import registry
from .base_model import BaseModel
from .blip_vision_t5_decoder import BlipVisionT5DecoderForConditionalGeneration, build_vision_projector
from .flux_prompt import FluxPipelineRewritePrompt
from .flux_transformer import FluxTransformer2DModel, FluxTransformerConfig
from .mllama_vllm_t5_embed_decoder_2 import MllamaVllmT5EmbedDecoderForConditionalGeneration_5
from .mllama_vllm_generate_1 import MllamaVllmGenerate_1
from .qwen2_vl import Qwen2VLTextConfig, Qwen2VLTextEngine, SamplingParams

__all__ = ["registry", "BaseModel", "BlipVisionT5DecoderForConditionalGeneration", "build_vision_projector",
           "FluxPipelineRewritePrompt", "FluxTransformer2DModel", "FluxTransformerConfig"]
