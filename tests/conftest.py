import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the tiny test shapes are (deliberately) not in the shipped MIOpen find-db
    config.addinivalue_line("filterwarnings", "ignore:pigeon_amd. the convolutions of UNet batch:UserWarning")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
