// api_gather.hip -- the one collective of the path: the optional final gather of the batch-sharded results over xGMI
// (north_star: "RCCL over xGMI only for the final gather").  Signals are independent, so the data path
// itself has no collective; an FFI consumer that wants every rank to hold all `Tx` shards calls this after its
// plan exec.  RCCL (librccl.so = NCCL's API on ROCm) is dlopen'd on first use: the library has no link-time dependency
// on it, and every entry point fails with a clear message when it is absent.  No torch anywhere.
#include <dlfcn.h>
#include <cstring>

#include <mutex>
#include <string>

#include "../../include/ssq_hip.h"
#include "ssq_common.h"

using namespace ssq;

namespace {

// the slice of rccl.h this file needs (ABI-stable NCCL 2.x signatures; /opt/rocm/include/rccl/rccl.h:40-43, :187,
// :220, :260 and the collectives section)
struct NcclUniqueId {
  char internal[128];
};
typedef void* NcclComm;
typedef int NcclResult;                // ncclSuccess == 0
constexpr int kNcclInt8 = 0;           // ncclDataType_t: ncclInt8 = ncclChar = 0

struct Rccl {
  void* handle = nullptr;
  NcclResult (*GetUniqueId)(NcclUniqueId*) = nullptr;
  NcclResult (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  NcclResult (*CommDestroy)(NcclComm) = nullptr;
  NcclResult (*CommCount)(const NcclComm, int*) = nullptr;
  NcclResult (*CommUserRank)(const NcclComm, int*) = nullptr;
  NcclResult (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(NcclResult) = nullptr;
  std::string why;                      // why loading failed
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
      r.why = dlerror() ? dlerror() : "dlopen failed";
    }
    if (!r.handle) {
      if (r.why.empty()) r.why = "librccl.so not found";
      return;
    }
    auto sym = [&](const char* s) {
      void* p = dlsym(r.handle, s);
      if (!p) r.why = std::string("librccl.so lacks ") + s;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.CommCount = (decltype(r.CommCount))sym("ncclCommCount");
    r.CommUserRank = (decltype(r.CommUserRank))sym("ncclCommUserRank");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.CommCount || !r.CommUserRank || !r.AllGather) {
      dlclose(r.handle);
      r.handle = nullptr;
    }
  });
  return r;
}

int need_rccl(Rccl** out) {
  Rccl& r = rccl();
  if (!r.handle) SSQ_FAIL("RCCL is not available: " + r.why);
  *out = &r;
  return 0;
}

#define SSQ_NCCL(r, call)                                                                                  \
  do {                                                                                                     \
    const NcclResult e__ = (call);                                                                         \
    if (e__ != 0) {                                                                                        \
      ::ssq::set_error(std::string(#call) + ": " + ((r)->GetErrorString ? (r)->GetErrorString(e__) : "RCCL error")); \
      return 3;                                                                                            \
    }                                                                                                      \
  } while (0)

}  // namespace

extern "C" {

int ssq_rccl_available(void) { return rccl().handle ? 1 : 0; }

int ssq_rccl_unique_id(void* id128) {
  if (!id128) SSQ_FAIL("id128 is NULL");
  Rccl* r = nullptr;
  if (int rc = need_rccl(&r)) return rc;
  {
    // (RCCL itself aborts its bootstrap with a "[FATAL ERROR]" line on stderr when no device is visible: say it ourselves)
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) SSQ_FAIL("no ROCm device is visible: RCCL needs one");
  }
  NcclUniqueId id;
  SSQ_NCCL(r, r->GetUniqueId(&id));
  std::memcpy(id128, id.internal, sizeof(id.internal));
  return 0;
}

int ssq_rccl_comm_init(void** comm, int n_ranks, const void* id128, int rank) {
  if (!comm || !id128) SSQ_FAIL("comm or id128 is NULL");
  *comm = nullptr;
  if (n_ranks < 1 || rank < 0 || rank >= n_ranks) SSQ_FAIL("bad rank / n_ranks");
  Rccl* r = nullptr;
  if (int rc = need_rccl(&r)) return rc;
  NcclUniqueId id;
  std::memcpy(id.internal, id128, sizeof(id.internal));
  NcclComm c = nullptr;
  SSQ_NCCL(r, r->CommInitRank(&c, n_ranks, id, rank));       // on the calling thread's current HIP device
  *comm = c;
  return 0;
}

int ssq_rccl_comm_destroy(void* comm) {
  if (!comm) return 0;
  Rccl* r = nullptr;
  if (int rc = need_rccl(&r)) return rc;
  SSQ_NCCL(r, r->CommDestroy((NcclComm)comm));
  return 0;
}

int ssq_rccl_comm_info(void* comm, int* n_ranks, int* rank) {
  if (!comm) SSQ_FAIL("comm is NULL");
  Rccl* r = nullptr;
  if (int rc = need_rccl(&r)) return rc;
  int n = 0, me = 0;
  SSQ_NCCL(r, r->CommCount((NcclComm)comm, &n));
  SSQ_NCCL(r, r->CommUserRank((NcclComm)comm, &me));
  if (n_ranks) *n_ranks = n;
  if (rank) *rank = me;
  return 0;
}

int ssq_gather_shards(void* comm, const void* d_send, void* d_recv, int64_t bytes_per_rank, void* stream) {
  if (!comm) SSQ_FAIL("comm is NULL");
  if (bytes_per_rank < 0) SSQ_FAIL("bytes_per_rank is negative");
  if (bytes_per_rank == 0) return 0;
  if (!d_send || !d_recv) SSQ_FAIL("device pointer is NULL");
  Rccl* r = nullptr;
  if (int rc = need_rccl(&r)) return rc;
  // shards are opaque bytes here: [rank][bytes_per_rank] in d_recv, rank order = batch order (batch.py::shard_bounds)
  SSQ_NCCL(r, r->AllGather(d_send, d_recv, (size_t)bytes_per_rank, kNcclInt8, (NcclComm)comm, (hipStream_t)stream));
  return 0;
}

}  // extern "C"
