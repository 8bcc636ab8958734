import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "gpu2: needs two MI355X on one node (RCCL); skipped elsewhere (run with -m gpu2)")


def pytest_collection_modifyitems(config, items):
    import torch
    ngpu = torch.cuda.device_count() if torch.cuda.is_available() else 0
    skip = pytest.mark.skip(reason="no GPU visible")
    skip2 = pytest.mark.skip(reason="needs 2 GPUs on this node (%d visible)" % ngpu)
    for item in items:
        if "gpu2" in item.keywords:
            if ngpu < 2:
                item.add_marker(skip2)
        elif "gpu" in item.keywords and ngpu < 1:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def gold():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLD, name + ".npz"))
    return load
