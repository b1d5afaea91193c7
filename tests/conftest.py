import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_ROOT = os.path.join(ROOT, "thinkdiff-mlre_amd")
for p in (ROOT, PKG_ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from thinkdiff import _hip
    _hip.lib()  # raises loudly if the HIP library was not built
    return _hip
