import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def native_lib():
    """The C-ABI library; built on demand here (hipcc cross-compiles without a GPU)."""
    from distributedfft_amd import _lib
    if not _lib.LIB_PATH.exists():
        from distributedfft_amd.build import build
        build()
    return _lib.load()


@pytest.fixture(scope="session")
def gpu(native_lib):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (the product path has no CPU fallback)")
    if native_lib.dfft_device_count() < 1:
        pytest.fail("libdfft_mi355x sees no HIP device")
    return torch.device("cuda:0")
