import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def golden_files(prefix):
    return sorted(f for f in os.listdir(GOLDEN)
                  if f.startswith(prefix) and f.endswith(".npz"))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _cpu_threads():
    """The GPU box has 256 hardware threads; the CPU oracle's per-step operands are small, and a pool that wide
    spends the time of a long recurrence in dispatch (C2 at T = 52 116: 294 s against 19 s).  16 threads serve both
    the small recurrences and the N = 100 000 sparse products of the tests."""
    import torch
    if torch.get_num_threads() > 16:
        torch.set_num_threads(16)
    yield
