import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def _guard_check(request):
    """S2S_GUARD=1|2 (development allocator mode, include/s2s_mi355x_dev.h): after every GPU test the poisoned gaps behind the engines'
    workspace buffers are verified - a kernel that stored past the end of a buffer fails the test that ran it, with the owner named."""
    yield
    if os.environ.get("S2S_GUARD", "0") in ("", "0") or "gpu" not in request.keywords or not _has_gpu():
        return
    from speech_to_speech_amd import native

    n, report = native.guard_check()
    assert n == 0, "S2S_GUARD: stores behind the end of a workspace buffer:\n" + report
