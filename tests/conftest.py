import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_addoption(parser):
    parser.addoption("--require-real-diffusers", action="store_true", default=False,
                     help="fail the golden-vector tests unless the goldens were made over a real `diffusers` (tests/golden/README.md)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the slowest tests in every summary: the GPU suite is dominated by the fp32 CPU oracle at full size, and the driver
    # gives it a fixed wall-clock budget
    if getattr(config.option, "durations", None) is None:
        config.option.durations = 10
        config.option.durations_min = 5.0


def rel_inf(a, b):
    """max|a-b| / max|b| per tensor -- the parity metric of BASELINE.md section 3."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


@pytest.fixture(scope="session", autouse=True)
def _oracle_threads():
    """the fp32 PyTorch oracle is fastest on <= 32 host threads (bench.py's cpu_baseline leg measured it: more threads
    oversubscribe the memory system of the GPU box); the GPU suite spends most of its wall time in it"""
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    yield


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
