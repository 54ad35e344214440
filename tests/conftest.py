import os
import sys

# the test session is an entry point: run the kernels the way bench.py does (kernel arguments staged in device memory); must
# happen before anything imports torch (missm_benchmark_amd/__init__.py); an explicit setting in the environment wins
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import torch
    return torch.load(os.path.join(GOLDEN, name + ".pt"), map_location="cpu", weights_only=False)
