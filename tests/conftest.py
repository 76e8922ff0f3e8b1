import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    config.addinivalue_line("markers", "slow: long-running CPU test (full-size oracle)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


def poison_gpu_memory(pattern=0xFFFFFFFF, big_gib=24, small_blocks=2048):
    """Fill the caching allocator's pools with a bit pattern and hand the blocks back: every later torch.empty (outputs, workspaces,
    id lists) then starts from that pattern instead of the zeros of fresh VRAM.  A kernel that reads memory nobody wrote -- and, worse,
    uses it as an index -- shows up as a wrong result or a GPU fault instead of passing by luck.  0xFFFFFFFF = NaN / -1, 0x7F7F7F7F =
    3.4e38 / 2139062143."""
    import torch
    if not torch.cuda.is_available():
        return
    val = pattern - (1 << 32) if pattern >= (1 << 31) else pattern
    big = [torch.full((2 ** 28,), val, dtype=torch.int32, device="cuda") for _ in range(big_gib)]          # 1 GiB each: large pool
    mid = [torch.full((2 ** 20 + 64 * k,), val, dtype=torch.int32, device="cuda") for k in range(256)]      # 4 MiB-ish
    small = [torch.full((32 * (1 + k % 512),), val, dtype=torch.int32, device="cuda") for k in range(small_blocks)]   # small pool
    torch.cuda.synchronize()
    del big, mid, small


@pytest.fixture(autouse=True)
def _poison_between_tests(request):
    """LOFTR_TEST_POISON=<hex pattern>: re-poison the allocator's free blocks before every GPU test (tools/gpu/r3_poison.sh)."""
    pat = os.environ.get("LOFTR_TEST_POISON")
    if pat and request.node.get_closest_marker("gpu"):
        poison_gpu_memory(int(pat, 16), big_gib=int(os.environ.get("LOFTR_TEST_POISON_GIB", "12")))
    yield
