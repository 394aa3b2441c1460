import os
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def nlc():
    """The product package, on a GPU box (the -m gpu tests): a missing device is a failure, not a skip."""
    import torch

    import neurallaplacecontrol_amd as n

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return n
