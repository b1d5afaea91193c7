"""`Config(args)` for the unchanged reference YAMLs (reference thinkdiff/common/config.py:16-169).

The reference merges, with OmegaConf: the `run` node, the model's default YAML (looked up through
`PRETRAINED_MODEL_CONFIG_DICT[model_type]`) under the user `model` node, each dataset's default YAML
under the user `datasets` node, `evaluation_datasets`, and `--options k=v` overrides.  OmegaConf is
not available here, so `Node` reproduces the access patterns the drivers use: attribute and item
access, `.get(k, default)`, plain `list` for sequences (drivers test `type(x) == list`).

Deliberate difference: the committed CLIP YAMLs list `datasets.laion`, whose builder the reference has
commented out, so the reference's own loader dereferences None (common/config.py:99-104).  Inference
needs no dataset builder: unknown datasets are kept as written instead of crashing.
"""
import json
import os

import yaml

from .registry import registry


class Node(dict):
    """dict with attribute access; nested dicts become Nodes, sequences become lists."""

    def __init__(self, data=None):
        super().__init__()
        for k, v in (data or {}).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, Node):
            return v
        if isinstance(v, dict):
            return Node(v)
        if isinstance(v, (list, tuple)):
            return [Node._wrap(x) for x in v]
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, Node._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Node) else v) for k, v in self.items()}


def merge(*nodes):
    """OmegaConf.merge semantics for mappings: later wins, dicts merge recursively, lists replace."""
    out = Node()
    for n in nodes:
        for k, v in (n or {}).items():
            if isinstance(v, dict) and isinstance(out.get(k), dict):
                out[k] = merge(out[k], v)
            else:
                out[k] = v
    return out


def _parse_scalar(s):
    try:
        return yaml.safe_load(s)
    except yaml.YAMLError:
        return s


def from_dotlist(opts):
    """['run.seed=1', 'model.ckpt', '/x'] or k=v forms (reference _convert_to_dot_list, config.py:171-185)."""
    opts = list(opts or [])
    if opts and not any("=" in o for o in opts):
        opts = [f"{k}={v}" for k, v in zip(opts[0::2], opts[1::2])]
    out = Node()
    for o in opts:
        k, v = o.split("=", 1)
        cur = out
        parts = k.split(".")
        for p in parts[:-1]:
            if p not in cur:
                cur[p] = Node()
            cur = cur[p]
        cur[parts[-1]] = _parse_scalar(v)
    return out


def load_yaml(path):
    with open(path) as fh:
        return Node(yaml.safe_load(fh) or {})


class Config:
    def __init__(self, args):
        self.config = Node()
        self.args = args
        registry.register("configuration", self)
        user = from_dotlist(getattr(args, "options", None))
        cfg = load_yaml(args.cfg_path)
        run = merge(Node({"run": cfg.get("run", {})}), Node({"run": user.get("run", {})}))
        model = self.build_model_config(cfg, user)
        datasets = self.build_dataset_config(cfg, "datasets")
        evals = self.build_dataset_config(cfg, "evaluation_datasets")
        self.config = merge(run, model, datasets, evals, Node({k: v for k, v in user.items() if k not in ("run", "model")}))

    @staticmethod
    def build_model_config(cfg, user):
        model = cfg.get("model", None)
        assert model is not None, "Missing model configuration file."
        model_cls = registry.get_model_class(model.get("arch"))
        assert model_cls is not None, f"Model '{model.get('arch')}' has not been registered."
        model_type = user.get("model", {}).get("model_type", None) or model.get("model_type", None)
        assert model_type is not None, "Missing model_type."
        default = Node()
        path = model_cls.default_config_path(model_type=model_type) if hasattr(model_cls, "default_config_path") else None
        if path and os.path.exists(path):
            default = load_yaml(path)
        return merge(default, Node({"model": model}), Node({"model": user.get("model", {})}))

    @staticmethod
    def build_dataset_config(cfg, key):
        ds = cfg.get(key, None)
        if ds is None:
            return Node()
        out = Node({key: Node()})
        for name in ds:
            builder = registry.get_builder_class(name)
            default = Node()
            if builder is not None and hasattr(builder, "default_config_path"):
                p = builder.default_config_path(type=ds[name].get("type", "default"))
                if p and os.path.exists(p):
                    default = load_yaml(p).get("datasets", Node())
            out[key] = merge(out[key], default, Node({name: ds[name]}))
        return out

    @property
    def run_cfg(self):
        return self.config.run

    @property
    def datasets_cfg(self):
        return self.config.get("datasets", Node())

    @property
    def evaluation_datasets_cfg(self):
        return self.config.get("evaluation_datasets", Node())

    @property
    def model_cfg(self):
        return self.config.model

    def get_config(self):
        return self.config

    def pretty_print(self):
        print("\n=====  Running Parameters    =====")
        print(json.dumps(self.config.run.to_dict(), indent=4, sort_keys=True))
        print("\n======  Model Attributes  ======")
        print(json.dumps(self.config.model.to_dict(), indent=4, sort_keys=True))

    def to_dict(self):
        return self.config.to_dict()
