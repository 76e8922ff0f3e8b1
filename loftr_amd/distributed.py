"""Multi-GPU data-parallel matching: one process per GPU, image pairs sharded by rank.

The reference does exactly this with Lightning DDP + DistributedSampler (src/lightning/data.py:315,
test.py:65) and merges results afterwards by a pickled gloo ``gather`` (src/utils/comm.py:179-219).
Here the only exchange on the data path is an all-gather of the per-pair match counts
(``int32[N_local]`` per rank -> ``int32[N_global]``): a few hundred bytes, latency bound, RCCL over
xGMI on GPUs (backend "nccl") and gloo on CPU for the tests.  With the counts every rank can rebase
its local ``b_ids`` into the global batch and knows the global offset of its matches; the optional
``all_gather_matches`` pads to the maximum count like comm.py:113-138 pads its byte tensors.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


def shard_bounds(n_items, rank, world):
    """Contiguous [lo, hi) slice of `n_items` pairs owned by `rank` (first ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class RcclCounts:
    """The data-path collective through the C-ABI (include/loftr_hip.h: loftr_rccl_*): an RCCL communicator owned by
    the library and one ``ncclAllGather`` of int32 counts on the current HIP stream -- no torch.distributed call on
    the data path.  torch.distributed (any backend) is only the control plane that carries rank 0's 128-byte
    unique id to the other ranks, as a DDP launcher's rendezvous store would.

    Raises LoftrHipError when RCCL cannot be set up (no fallback inside; callers decide -- bench.py records which
    transport ran)."""

    def __init__(self, device, group=None):
        if not dist.is_initialized():
            raise _lib.LoftrHipError("RcclCounts needs an initialised torch.distributed process group (control plane)")
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        nbytes = 128                                           # LOFTR_RCCL_ID_BYTES
        buf = C.create_string_buffer(nbytes)
        if self.rank == 0:
            _lib.check(self.lib.loftr_rccl_unique_id(buf, nbytes), "loftr_rccl_unique_id")
        box = [bytes(buf.raw) if self.rank == 0 else None]
        # `src` of a broadcast is a GLOBAL rank: the id comes from the process that is rank 0 OF THIS GROUP (a sub-group need not
        # contain global rank 0)
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):                    # the communicator binds to the current device
            _lib.check(self.lib.loftr_rccl_comm_create(box[0], nbytes, self.rank, self.world, C.byref(handle)),
                       "loftr_rccl_comm_create")
        self.comm = handle
        r, w = C.c_int(-1), C.c_int(-1)
        _lib.check(self.lib.loftr_rccl_comm_info(self.comm, C.byref(r), C.byref(w)), "loftr_rccl_comm_info")
        assert (r.value, w.value) == (self.rank, self.world)
        self.ranks_seen = w.value

    def all_gather(self, counts):
        """int32 [n] on this rank's GPU -> int32 [world * n] in rank order (same n on every rank)."""
        if counts.dtype != torch.int32 or not counts.is_cuda or not counts.is_contiguous():
            raise _lib.LoftrHipError("RcclCounts.all_gather: expected a contiguous int32 GPU tensor")
        out = torch.empty(self.world * counts.numel(), dtype=torch.int32, device=counts.device)
        with torch.cuda.device(counts.device):
            st = torch.cuda.current_stream(counts.device).cuda_stream
            _lib.check(self.lib.loftr_rccl_allgather_counts(self.comm, C.c_void_p(counts.data_ptr()),
                                                            C.c_void_p(out.data_ptr()), counts.numel(), C.c_void_p(st)),
                       "loftr_rccl_allgather_counts")
        return out

    def close(self):
        if getattr(self, "comm", None):
            self.lib.loftr_rccl_comm_destroy(self.comm)
            self.comm = None


def all_gather_match_counts(local_counts, n_global, group=None, rccl=None):
    """local_counts int32 [N_local] (data['_match_counts'][1:]) -> int32 [n_global] in rank order.

    Ranks may own different numbers of pairs (shard_bounds); shorter shards are padded for the
    collective and the padding removed afterwards.  ``rccl``: an RcclCounts -> the collective runs through the
    library's C-ABI; otherwise through torch.distributed (gloo on CPU in the tests, RCCL under backend "nccl")."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n_max = -(-n_global // world)
    buf = torch.zeros(n_max, dtype=torch.int32, device=local_counts.device)
    buf[: local_counts.numel()] = local_counts.to(torch.int32)
    if rccl is not None:
        out = rccl.all_gather(buf)
    else:
        out = torch.empty(world * n_max, dtype=torch.int32, device=local_counts.device)
        dist.all_gather_into_tensor(out, buf, group=group)
    parts = []
    for r in range(world):
        lo, hi = shard_bounds(n_global, r, world)
        parts.append(out[r * n_max: r * n_max + (hi - lo)])
    del rank
    return torch.cat(parts)


def globalize(data, n_global, group=None, rccl=None):
    """Adds the global view to a rank-local batch dict after forward():
    ``match_counts_global`` [n_global], ``b_ids_global`` (local b_ids rebased by the shard offset)
    and ``match_offset`` (index of this rank's first match in the concatenated global match list)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = all_gather_match_counts(data["_match_counts"][1:], n_global, group, rccl)
    lo, _ = shard_bounds(n_global, rank, world)
    data["match_counts_global"] = counts
    data["b_ids_global"] = data["b_ids"] + lo
    data["match_offset"] = int(counts[:lo].sum().item()) if lo else 0
    return data


def all_gather_matches(data, group=None):
    """[M_global, 5] float32 rows (mkpts0_f, mkpts1_f, mconf) + global b_ids, in ascending global
    (b, i) order.  Padded all-gather (20 B per match); optional -- not on the bench's timed path."""
    world = dist.get_world_size(group)
    counts = data["match_counts_global"]
    rows = torch.cat([data["mkpts0_f"], data["mkpts1_f"], data["mconf"][:, None],
                      data["b_ids_global"].to(torch.float32)[:, None]], 1)
    m_local = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    m_all = [torch.zeros_like(m_local) for _ in range(world)]
    dist.all_gather(m_all, m_local, group=group)
    m_all = [int(m.item()) for m in m_all]
    m_max = max(max(m_all), 1)
    pad = torch.zeros(m_max, 6, dtype=torch.float32, device=rows.device)
    pad[: rows.shape[0]] = rows
    out = torch.empty(world * m_max, 6, dtype=torch.float32, device=rows.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    parts = [out[r * m_max: r * m_max + m_all[r]] for r in range(world)]
    allrows = torch.cat(parts)
    assert int(counts.sum().item()) == allrows.shape[0]
    return allrows[:, :5], allrows[:, 5].to(torch.int64)
