import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    if os.environ.get("LSSVR_HIP_LIB"):
        raise pytest.UsageError("LSSVR_HIP_LIB is set: the test suite only runs against the in-tree "
                                "hybrid_fem_lssvr_amd/csrc/liblssvr_hip.so")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU is visible: the HIP path has no CPU fallback")
    return torch.device("cuda:0")
