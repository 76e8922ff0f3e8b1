"""GPU test of the C-ABI collective (include/loftr_hip.h: loftr_rccl_*): communicator creation from a unique id and
the int32 count all-gather on a HIP stream.  A gpurun box has ONE GPU, so this is the world-size-1 path (RCCL refuses
two ranks on one device); ranks > 1 share every line of code with it except the communicator size and are covered by
`tools/rccl_world.py` (any N visible GPUs) and by the driver's multi-GPU bench.  The gloo world-2 tests in
test_distributed_gloo.py cover the host logic (padding, shard bounds, rebasing)."""
import os
import tempfile

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def test_rccl_counts_world1():
    from loftr_amd.distributed import RcclCounts, all_gather_match_counts
    assert not dist.is_initialized()
    with tempfile.TemporaryDirectory() as d:
        dist.init_process_group("gloo", init_method=f"file://{os.path.join(d, 'rdv')}", rank=0, world_size=1)
        try:
            dev = torch.device("cuda", 0)
            rc = RcclCounts(dev)
            assert rc.ranks_seen == 1
            counts = torch.tensor([5, 0, 17, 3, 900, 1, 2, 44], dtype=torch.int32, device=dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):                       # the collective follows the caller's current stream
                out = rc.all_gather(counts)
            side.synchronize()
            assert out.tolist() == counts.tolist()
            out2 = all_gather_match_counts(counts[:5], 5, rccl=rc)     # ragged path: padding + trimming
            torch.cuda.synchronize()
            assert out2.tolist() == counts[:5].tolist()
            rc.close()
        finally:
            dist.destroy_process_group()
