import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    # Let torch bring the HIP runtime up before the engine library does: a test that imports torch after the
    # library has created its streams has been seen to get "No HIP GPUs are available" from torch on the GPU box.
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def ora():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
