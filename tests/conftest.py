"""Shared pytest configuration.

* ``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI symbol check -- no GPU needed.
* ``-m gpu``: parity of the HIP path (through the C ABI) against the oracle on a real MI355X.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")


@pytest.fixture(scope="session")
def hot():
    """The device engine; GPU tests fail loudly (not skip) when the extension cannot be loaded."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device in this process")
    from marex_amd.csrc import build as _b

    _b.build(verbose=False)
    from marex_amd.detect import get_engine
    from marex_amd.engine import HotPath

    HotPath.POISON = True  # fresh output buffers start as 0xCD bytes: unwritten elements cannot pass as zeros

    return get_engine(0)  # the engine the public API uses: options set on it in a test reach preprocess_data too
