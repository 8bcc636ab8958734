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


# Collection order (the driver runs `pytest -m gpu -x`): the reference-golden and oracle parity tests first -- they are cheap,
# deterministic and carry the parity claim --, kernel unit tests next, the full-size oracle comparisons after that, the long
# sampler runs last, so that a late failure can never hide the golden comparisons.
_FILE_ORDER = ["test_oracle_golden", "test_thirdparty_pins", "test_host_cpu", "test_parity_gpu", "test_mid_gpu", "test_kernels_gpu", "test_x3_gpu",
               "test_harness_gpu", "test_conv3_gpu", "test_attention_gpu", "test_determinism_gpu", "test_fullsize_gpu", "test_configs_b16_gpu",
               "test_multigpu"]
_LATE_TESTS = ("test_e2e_", "test_baseline_config_shapes_run")


def _order_key(item):
    mod = item.module.__name__.rsplit(".", 1)[-1] if item.module is not None else ""
    rank = _FILE_ORDER.index(mod) if mod in _FILE_ORDER else len(_FILE_ORDER)
    late = 1 if item.name.startswith(_LATE_TESTS) else 0
    return (late, rank)


def pytest_collection_modifyitems(config, items):
    import torch
    items.sort(key=_order_key)          # stable: definition order kept inside a file
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
