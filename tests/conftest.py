import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    # the CPU oracle (torch / oneDNN) is several times SLOWER with one thread per hardware thread on the 256-thread GPU
    # hosts than with ~16 (bench.py's thread sweep): cap the default for every test
    import os

    import torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
