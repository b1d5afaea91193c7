"""Runner shims (reference thinkdiff/runners/runner_base.py:46-96, runner_clip_t5.py): the inference drivers
instantiate a runner only to reach `runner.model` (device-placed, DDP-like `.module`)."""
from ..common.registry import registry
from . import dp_inference


class RunnerBase:
    def __init__(self, cfg, task, model, datasets, job_id):
        self.config, self.task, self.datasets, self.job_id = cfg, task, datasets, job_id
        self._model = model

    @property
    def device(self):
        return self.config.run_cfg.get("device", "cuda")

    @property
    def model(self):
        """Already resident on the GPU; `.module` aliases the model itself (no DDP wrap at inference)."""
        return self._model.to(self.device) if hasattr(self._model, "to") else self._model

    def train(self):
        raise NotImplementedError("training is outside the MI355X inference hot path (SURVEY.md 8)")


@registry.register_runner("runner_base")
class _RunnerBase(RunnerBase):
    pass


@registry.register_runner("runner_clip_t5")
class RunnerClipT5(RunnerBase):
    pass


from .runner_process_data import RunnerProcessData  # noqa: E402

__all__ = ["RunnerBase", "RunnerClipT5", "RunnerProcessData", "dp_inference"]
