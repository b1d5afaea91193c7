"""torch.ops.thinkdiff_hip.* (SURVEY.md 8(b) custom-op layer): lib/libthinkdiff_torch_ops.so loads without a GPU, registers the
documented schemas, and has no host kernel to fall back to."""
import pytest
import torch


def test_namespace_and_schemas():
    import thinkdiff.ops as ops
    import os
    assert ops.register() is ops.register() and os.path.exists(ops.OPS_LIB_PATH)
    for name, sig in ops.SCHEMAS.items():
        op = getattr(torch.ops.thinkdiff_hip, name)
        assert str(op.default._schema) == f"thinkdiff_hip::{name}{sig}"
    s = str(torch.ops.thinkdiff_hip.linear.default._schema)
    assert "Tensor? bias" in s and "int act" in s and s.endswith("-> Tensor")
    assert "Tensor(a!) x" in str(torch.ops.thinkdiff_hip.euler_step_.default._schema)


def test_no_cpu_kernel():
    import thinkdiff.ops  # noqa: F401
    x, w = torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises((NotImplementedError, RuntimeError)):
        torch.ops.thinkdiff_hip.linear(x, w, None, 0, None, None)
