"""BaseModel glue with the reference's surface (thinkdiff/models/base_model.py:30-111)."""
import os

import torch

from ..common.config import load_yaml


class BaseModel:
    PRETRAINED_MODEL_CONFIG_DICT = {}

    @property
    def device(self):
        return getattr(self, "_device", torch.device("cuda" if torch.cuda.is_available() else "cpu"))

    @classmethod
    def default_config_path(cls, model_type):
        assert model_type in cls.PRETRAINED_MODEL_CONFIG_DICT, f"Unknown model type {model_type}"
        rel = cls.PRETRAINED_MODEL_CONFIG_DICT[model_type]
        return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), rel)

    @classmethod
    def from_pretrained(cls, model_type):
        return cls.from_config(load_yaml(cls.default_config_path(model_type)).model)

    def eval(self):
        return self

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def to(self, device=None, dtype=None):
        if device is not None:
            self._device = torch.device(device)
        return self

    @property
    def module(self):
        """DDP-style alias: LVLM drivers call `model.module.get_embed`."""
        return self
