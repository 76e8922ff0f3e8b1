#!/usr/bin/env python
"""N-rank check of the C-ABI count all-gather on N visible GPUs (one process per GPU):

    python tools/rccl_world.py [N]          # spawns N ranks itself (default: all visible GPUs)

Each rank builds an RcclCounts communicator (unique id from rank 0 over the torch.distributed store), gathers a
rank-dependent int32 vector through loftr_rccl_allgather_counts and compares with torch.distributed's own
all_gather_into_tensor.  Prints one OK line per rank."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker():
    import torch
    import torch.distributed as dist
    from loftr_amd.distributed import RcclCounts
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    rc = RcclCounts(dev)
    n = 8
    mine = (torch.arange(n, dtype=torch.int32, device=dev) + 100 * rank).contiguous()
    got = rc.all_gather(mine)
    want = torch.empty(world * n, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(want, mine)
    torch.cuda.synchronize()
    assert torch.equal(got, want), (rank, got.tolist(), want.tolist())


    print(f"rank {rank}/{world} on {torch.cuda.get_device_name(local)}: C-ABI RCCL all-gather OK "
          f"({rc.ranks_seen} ranks in the communicator)", flush=True)
    rc.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    if "WORLD_SIZE" in os.environ:
        worker()
    else:
        import torch
        n = int(sys.argv[1]) if len(sys.argv) > 1 else torch.cuda.device_count()
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)], env=env))
