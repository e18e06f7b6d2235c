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


@pytest.fixture(autouse=True)
def _poison_free_gpu_memory(request):
    """G2V_TEST_POISON=1: before every GPU test, fill the caching allocator's free blocks with NaN bit patterns (a fresh
    process hands out zero pages, which hides reads of memory a kernel was supposed to write first).  Any such read turns
    the test's numbers into NaN instead of passing by luck."""
    if os.environ.get("G2V_TEST_POISON") == "1" and request.node.get_closest_marker("gpu") is not None:
        import torch
        if torch.cuda.is_available():
            blocks = []
            try:
                for mb in (2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1):
                    for _ in range(2):
                        blocks.append(torch.full((mb * 1024 * 1024 // 4,), float("nan"), device="cuda:0"))
                blocks += [torch.full((n,), float("nan"), device="cuda:0") for n in (1 << 16, 1 << 14, 1 << 12, 1 << 10, 256, 64) for _ in range(8)]
            except RuntimeError:
                pass
            del blocks
    yield
