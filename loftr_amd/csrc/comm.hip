// Multi-GPU exchange of the matching path: all-gather of the per-pair match counts over RCCL / xGMI.
//
// The reference shards image pairs over GPUs with Lightning DDP + DistributedSampler (test.py:65,
// src/lightning/data.py:315) and merges results by a pickled gloo gather (src/utils/comm.py:113-219).  Here
// pairs are independent, so the only data-path collective is `int32[n]` per rank -> `int32[world * n]` in rank
// order (SURVEY.md §8(b),(e)): latency bound, one ncclAllGather on the caller's stream.
//
// librccl is opened lazily with dlopen so that the library (and its CPU-side ABI checks) load on boxes where no
// communicator is ever created; the communicator handle crosses the C-ABI as an opaque pointer.  The 128-byte
// unique id is produced by rank 0 and handed to the other ranks by the caller's own control plane (the Python
// layer uses the torch.distributed store it already has for rendezvous).
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>
#include "common.h"

namespace {

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.h) break;
    }
    if (!r.h) return;
    auto sym = [&](const char* n) { return dlsym(r.h, n); };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommCount && r.CommUserRank && r.AllGather;
  });
  return r;
}

}  // namespace

extern "C" int loftr_rccl_unique_id(char* id_out, size_t id_bytes) {
  LOFTR_CHECK_ARG(id_out && id_bytes >= LOFTR_RCCL_ID_BYTES);
  static_assert(sizeof(ncclUniqueId) == LOFTR_RCCL_ID_BYTES, "unique id size");
  Rccl& r = rccl();
  if (!r.ok) return LOFTR_ERR_COMM;
  ncclUniqueId id;
  if (r.GetUniqueId(&id) != ncclSuccess) return LOFTR_ERR_COMM;
  memcpy(id_out, &id, sizeof(id));
  return LOFTR_OK;
}

extern "C" int loftr_rccl_comm_create(const char* id, size_t id_bytes, int rank, int world, void** comm_out) {
  LOFTR_CHECK_ARG(id && id_bytes >= LOFTR_RCCL_ID_BYTES && comm_out && world >= 1 && rank >= 0 && rank < world);
  Rccl& r = rccl();
  if (!r.ok) return LOFTR_ERR_COMM;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  if (r.CommInitRank(&c, world, uid, rank) != ncclSuccess) return LOFTR_ERR_COMM;   // binds to the CURRENT HIP device
  *comm_out = (void*)c;
  return LOFTR_OK;
}

extern "C" int loftr_rccl_comm_info(void* comm, int* rank_out, int* world_out) {
  LOFTR_CHECK_ARG(comm && rank_out && world_out);
  Rccl& r = rccl();
  if (!r.ok) return LOFTR_ERR_COMM;
  if (r.CommUserRank((ncclComm_t)comm, rank_out) != ncclSuccess) return LOFTR_ERR_COMM;
  if (r.CommCount((ncclComm_t)comm, world_out) != ncclSuccess) return LOFTR_ERR_COMM;
  return LOFTR_OK;
}

extern "C" int loftr_rccl_comm_destroy(void* comm) {
  if (!comm) return LOFTR_OK;
  Rccl& r = rccl();
  if (!r.ok) return LOFTR_ERR_COMM;
  return r.CommDestroy((ncclComm_t)comm) == ncclSuccess ? LOFTR_OK : LOFTR_ERR_COMM;
}

extern "C" int loftr_rccl_allgather_counts(void* comm, const int32_t* counts_in, int32_t* counts_out, int n,
                                           void* stream) {
  LOFTR_CHECK_ARG(comm && counts_in && counts_out && n > 0);
  Rccl& r = rccl();
  if (!r.ok) return LOFTR_ERR_COMM;
  if (r.AllGather(counts_in, counts_out, (size_t)n, ncclInt32, (ncclComm_t)comm, (hipStream_t)stream) != ncclSuccess)
    return LOFTR_ERR_COMM;
  return LOFTR_OK;
}
