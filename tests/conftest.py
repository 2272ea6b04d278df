import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    """The product library bound to device 0.  Fails loudly (no CPU fallback) without a GPU."""
    import torch  # noqa: F401  first, so that the process holds ONE HIP runtime (torch bundles its own)
    import jpeg_decoder_amd as jg
    jg.init(0)
    return jg
