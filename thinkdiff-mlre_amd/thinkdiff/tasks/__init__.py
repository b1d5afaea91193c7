"""`tasks.setup_task(cfg)` (reference thinkdiff/tasks/__init__.py:13-20, base_task.py:30-35)."""
from ..common.registry import registry


class BaseTask:
    @classmethod
    def setup_task(cls, **kwargs):
        return cls()

    def build_model(self, cfg):
        model_config = cfg.model_cfg
        model_cls = registry.get_model_class(model_config.arch)
        return model_cls.from_config(model_config)

    def build_datasets(self, cfg):
        """Inference drivers build datasets only as a side effect; nothing to build here."""
        return {}


@registry.register_task("image_text_pretrain")
class ImageTextPretrainTask(BaseTask):
    pass


def setup_task(cfg):
    assert "task" in cfg.run_cfg, "Task name must be provided."
    task_name = cfg.run_cfg.task
    cls = registry.get_task_class(task_name)
    assert cls is not None, f"Task {task_name} not properly registered."
    return cls.setup_task(cfg=cfg)


from .image_text_process_data import ImageTextProcessDataTask  # noqa: E402,F401
