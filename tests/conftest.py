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


@pytest.fixture
def note(request):
    """Record a measured quantity next to its bar: appended to gpurun_out/measured_bars.log (merged
    back from the GPU box) so that bars can be pinned at a stated multiple of what was measured."""
    def _note(what, value, bar=None):
        d = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "measured_bars.log"), "a") as fh:
                fh.write("%s :: %s = %.3e%s\n" % (request.node.name, what, float(value),
                                                   "" if bar is None else "  (bar %.1e)" % bar))
        except OSError:
            pass
    return _note
