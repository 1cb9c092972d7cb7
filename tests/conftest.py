import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "willow-inference-server_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle's decoder passes are chains of skinny GEMVs: on a many-core host (the GPU box has 128 hardware threads) torch's
    # default of one thread per core spends its time in hand-offs (measured: the large-v2 oracle checks 338 s -> 95 s at 16 threads);
    # tests that want more for an encoder-sized GEMM set it themselves (tests/test_gpu_fullsize.py)
    try:
        import torch
        torch.set_num_threads(max(1, min(16, os.cpu_count() or 8)))
    except ImportError:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def lib():
    from wis_hip import _lib
    return _lib.load()
