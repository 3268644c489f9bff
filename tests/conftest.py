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


_POISON_BYTES = (0xFF, 0x7F)
_poison_turn = [0]


@pytest.fixture(autouse=True)
def _poisoned_device_memory(request):
    """Before every GPU test: hand the caching allocator's blocks back to the driver and leave ONE freshly poisoned 24 GiB block in the cache, so
    that whatever the test's first torch.empty() calls return (workspaces, packed weights, gradient buffers are all torch.empty) is NaN bytes
    (every other test: 0x7F bytes = 3.4e38), never the zeros of a fresh mapping or a plausible tensor of an earlier test.  A result that depends on
    what the allocator hands out is an engine bug; this makes such a read fail loudly and in every run.
    HN_TEST_POISON=0 switches the fixture off (A/B runs)."""
    if request.node.get_closest_marker("gpu") is None or os.environ.get("HN_TEST_POISON", "1") == "0":
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    byte = _POISON_BYTES[_poison_turn[0] % len(_POISON_BYTES)]
    _poison_turn[0] += 1
    blk = torch.empty(24 << 30, dtype=torch.uint8, device="cuda:0")
    blk.fill_(byte)
    torch.cuda.synchronize()
    del blk
    yield
