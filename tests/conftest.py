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


@pytest.fixture(autouse=True)
def _poison_device_memory(request):
    """Before every GPU test: fill the memory a kernel might read WITHOUT having written it with NaN bit patterns, so
    that such a read fails the test instead of passing on the zeros a fresh process happens to hold.
      * LDS is not cleared between kernels: an attention launch over NaN inputs stages NaN K / V tiles on every CU;
      * torch's caching allocator hands freed blocks back unchanged: NaN-filled blocks are freed into its large and
        small pools, so the next `torch.empty` of a test (outputs, workspaces) starts as NaN.
    Found with it: a kernel that multiplied unwritten LDS columns by zero weights (NaN only after kernels that leave
    NaN patterns in LDS, i.e. never in a fresh process)."""
    if "gpu" in request.keywords:
        import torch
        if torch.cuda.is_available():
            from pigeon_amd import ops
            nan = torch.full((64, 512, 640), float("nan"), device="cuda", dtype=torch.bfloat16)
            ops.attention(nan, nan, nan, 10)
            big = torch.full((3 << 28,), float("nan"), device="cuda", dtype=torch.float32)       # 3 GiB, large pool
            small = [torch.full((64 << 10,), float("nan"), device="cuda", dtype=torch.float32) for _ in range(64)]
            del nan, big, small
    yield
