// The one collective of the data-parallel step behind the C ABI (`evf_allreduce_sum`, 8(e)): a thin binding of
// RCCL's ncclAllReduce on a communicator of this library's own.  The reference has no distributed code at all
// (/root/reference/configs/parser.py:83-86: one process, num_workers = 0); its loss sums over the batch
// (loss/flow.py:226,259,289), so data parallelism is ONE in-place SUM all-reduce of the flat gradient buffer per optimizer
// step (299 KB for the FireNets: latency bound over xGMI).
//
// Why a communicator of our own and not torch.distributed's: torch issues a collective on a stream of its choosing and hands
// its end event to a watchdog thread -- neither can sit inside a hipGraph capture, so the N-rank step had to be TWO graphs with
// an eager all-reduce between them.  ncclAllReduce itself is capturable: on our communicator it is one more kernel node of the
// step's ONE graph, on the stream of every other launch.
//
// RCCL is bound at run time (dlopen / dlsym), not at link time: libevflow_hip.so loads on a box without RCCL, and inside a
// PyTorch process it binds the librccl that torch has already loaded (evf_comm_load(path) with torch/lib/librccl.so -- one
// copy of the library, one set of IPC / xGMI resources per process).  Every entry point fails loudly (EVF_ENOTSUP) when no
// RCCL is found.
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "evf_common.h"

namespace {
// The few RCCL / NCCL types and enumerators this file needs, declared here so that the BUILD needs no RCCL headers either (their
// values are part of NCCL's stable ABI: nccl.h `ncclResult_t`, `ncclUniqueId` = 128 opaque bytes, `ncclDataType_t`, `ncclRedOp_t`).
typedef int ncclResult_t;
constexpr ncclResult_t ncclSuccess = 0;
struct ncclUniqueId {
  char internal[128];
};
typedef struct ncclComm* ncclComm_t;
typedef int ncclDataType_t;
constexpr ncclDataType_t ncclFloat = 7;
typedef int ncclRedOp_t;
constexpr ncclRedOp_t ncclSum = 0, ncclMax = 2;

struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl rccl;
std::mutex rccl_mu;      // binding the library
std::mutex rccl_err_mu;  // the text of the last error (written from any entry point)
char rccl_err[256] = "";

void set_err(const char* a, const char* b) {
  std::lock_guard<std::mutex> g(rccl_err_mu);
  snprintf(rccl_err, sizeof rccl_err, "%s: %s", a, b ? b : "(no text)");
}

bool bind(void* h) {
  if (!h) return false;
  Rccl r;
  r.h = h;
  r.GetVersion = (decltype(r.GetVersion))dlsym(h, "ncclGetVersion");
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.CommCount = (decltype(r.CommCount))dlsym(h, "ncclCommCount");
  r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.GetVersion || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) return false;
  rccl = r;
  return true;
}

int load_locked(const char* path) {
  if (rccl.h) return EVF_OK;
  if (path && *path && bind(dlopen(path, RTLD_NOW | RTLD_LOCAL))) return EVF_OK;
  // a librccl this process has loaded already (PyTorch's), then the system one
  for (const char* name : {"librccl.so", "librccl.so.1"})
    if (bind(dlopen(name, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD))) return EVF_OK;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
    if (bind(dlopen(name, RTLD_NOW | RTLD_LOCAL))) return EVF_OK;
  const char* e = dlerror();  // (read ONCE: dlerror() clears the message it returns)
  set_err("no usable librccl", e ? e : "symbols missing");
  return EVF_ENOTSUP;
}

int nccl_rc(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return EVF_OK;
  set_err(what, rccl.GetErrorString ? rccl.GetErrorString(r) : "RCCL error");
  return -(2000 + (int)r);
}
}  // namespace

extern "C" int evf_comm_load(const char* librccl_path) {
  std::lock_guard<std::mutex> g(rccl_mu);
  return load_locked(librccl_path);
}

extern "C" const char* evf_comm_last_error() { return rccl_err; }

extern "C" int evf_comm_version(int* version) {
  std::lock_guard<std::mutex> g(rccl_mu);
  if (!version) return EVF_EINVAL;
  if (const int rc = load_locked(nullptr)) return rc;
  return nccl_rc(rccl.GetVersion(version), "ncclGetVersion");
}

extern "C" int evf_comm_unique_id(void* id128) {
  std::lock_guard<std::mutex> g(rccl_mu);
  if (!id128) return EVF_EINVAL;
  if (const int rc = load_locked(nullptr)) return rc;
  static_assert(sizeof(ncclUniqueId) == EVF_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  return nccl_rc(rccl.GetUniqueId((ncclUniqueId*)id128), "ncclGetUniqueId");
}

extern "C" int evf_comm_init(const void* id128, int rank, int world, void** comm) {
  if (!id128 || !comm || world <= 0 || rank < 0 || rank >= world) return EVF_EINVAL;
  {
    std::lock_guard<std::mutex> g(rccl_mu);
    if (const int rc = load_locked(nullptr)) return rc;
  }
  ncclUniqueId id;
  memcpy(&id, id128, sizeof id);
  ncclComm_t c = nullptr;
  const int rc = nccl_rc(rccl.CommInitRank(&c, world, id, rank), "ncclCommInitRank");  // (blocks until every rank has called it)
  if (rc) return rc;
  *comm = (void*)c;
  return EVF_OK;
}

// Number of ranks RCCL itself sees on `comm` (ncclCommCount): the bench line records it, so that a scaling record shows
// whether the collective really ran over N ranks.
extern "C" int evf_comm_count(void* comm, int* ranks) {
  if (!comm || !ranks) return EVF_EINVAL;
  if (!rccl.h || !rccl.CommCount) return EVF_ENOTSUP;
  return nccl_rc(rccl.CommCount((ncclComm_t)comm, ranks), "ncclCommCount");
}

extern "C" int evf_comm_destroy(void* comm) {
  if (!comm) return EVF_OK;
  if (!rccl.h) return EVF_ENOTSUP;
  return nccl_rc(rccl.CommDestroy((ncclComm_t)comm), "ncclCommDestroy");
}

// In-place SUM over the ranks of `comm` of n floats at `buf`, enqueued on `stream` (capturable: one kernel node).
extern "C" int evf_allreduce_sum(void* comm, float* buf, int64_t n, void* stream) {
  if (!comm || !buf || n <= 0) return EVF_EINVAL;
  if (!rccl.h) return EVF_ENOTSUP;
  return nccl_rc(rccl.AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, (ncclComm_t)comm, EVF_STREAM(stream)), "ncclAllReduce");
}

// ... and MAX (the loader flags every rank has to act on together, the max-over-ranks clock of bench.py)
extern "C" int evf_allreduce_max(void* comm, float* buf, int64_t n, void* stream) {
  if (!comm || !buf || n <= 0) return EVF_EINVAL;
  if (!rccl.h) return EVF_ENOTSUP;
  return nccl_rc(rccl.AllReduce(buf, buf, (size_t)n, ncclFloat, ncclMax, (ncclComm_t)comm, EVF_STREAM(stream)), "ncclAllReduce(max)");
}
