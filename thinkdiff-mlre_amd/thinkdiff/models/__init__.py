"""Model registry (reference thinkdiff/models/__init__.py): importing this package registers the archs."""
from ..common.registry import registry
from .base_model import BaseModel
from .blip_vision_t5_decoder import BlipVisionT5DecoderForConditionalGeneration, build_vision_projector
from .flux_prompt import FluxPipelineRewritePrompt
from .flux_transformer import FluxTransformer2DModel, FluxTransformerConfig

__all__ = ["registry", "BaseModel", "BlipVisionT5DecoderForConditionalGeneration", "build_vision_projector",
           "FluxPipelineRewritePrompt", "FluxTransformer2DModel", "FluxTransformerConfig"]
