// Gradient exchange over RCCL (the one exchange step on the path: SURVEY.md 8e; reference training_script.py:190-199 steps
// a single process, its DDP analogue sums gradients over ranks).  C-ABI entries of include/msmd_hip.h:
//   msmd_comm_unique_id / msmd_comm_init / msmd_comm_destroy    one communicator per process (one process per GPU)
//   msmd_allreduce_bucket                                        in-place SUM of one gradient bucket on the caller's stream
// librccl is resolved at FIRST USE: a process that never exchanges (inference, one-GPU training) does not load it.  A copy the
// process has ALREADY mapped (torch's bundled librccl, by soname) is adopted first with RTLD_NOLOAD so that one process never
// holds two RCCL runtimes; only then is the soname / the ROCm path loaded.  The library's major version is checked against the
// slice of rccl.h restated below (ncclGetVersion: 2.x), anything else is refused with 1003.
#include "common.h"
#include <dlfcn.h>
#include <cstring>
#include <mutex>

namespace {
// the slice of rccl.h this file needs (types by value: ncclUniqueId = 128 opaque bytes, enums as int)
struct UniqueId { char internal[128]; };
typedef int (*get_unique_id_t)(UniqueId*);
typedef int (*comm_init_rank_t)(void**, int, UniqueId, int);
typedef int (*comm_destroy_t)(void*);
typedef int (*all_reduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*get_error_string_t)(int);
typedef int (*comm_count_t)(void*, int*);
typedef int (*get_version_t)(int*);

struct Rccl {
  void* handle = nullptr;
  get_unique_id_t get_unique_id = nullptr;
  comm_init_rank_t comm_init_rank = nullptr;
  comm_destroy_t comm_destroy = nullptr;
  all_reduce_t all_reduce = nullptr;
  comm_count_t comm_count = nullptr;
  int version = 0;   // ncclGetVersion: major * 10000 + minor * 100 + patch
  int status = -1;   // 0 = resolved, 1 = not found, 2 = symbols missing, 3 = unknown major version
};
Rccl g_rccl;
std::once_flag g_once;

const Rccl& rccl() {
  std::call_once(g_once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {          // a copy that is already mapped (torch's) wins: never two runtimes in one process
      g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      if (g_rccl.handle) break;
    }
    for (const char* n : names) {
      if (g_rccl.handle) break;
      g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!g_rccl.handle) { g_rccl.status = 1; return; }
    get_version_t get_version = (get_version_t)dlsym(g_rccl.handle, "ncclGetVersion");
    int version = 0;
    if (!get_version || get_version(&version) != 0 || version / 10000 != 2) { g_rccl.status = 3; return; }
    g_rccl.version = version;
    g_rccl.get_unique_id = (get_unique_id_t)dlsym(g_rccl.handle, "ncclGetUniqueId");
    g_rccl.comm_init_rank = (comm_init_rank_t)dlsym(g_rccl.handle, "ncclCommInitRank");
    g_rccl.comm_destroy = (comm_destroy_t)dlsym(g_rccl.handle, "ncclCommDestroy");
    g_rccl.all_reduce = (all_reduce_t)dlsym(g_rccl.handle, "ncclAllReduce");
    g_rccl.comm_count = (comm_count_t)dlsym(g_rccl.handle, "ncclCommCount");
    g_rccl.status = (g_rccl.get_unique_id && g_rccl.comm_init_rank && g_rccl.comm_destroy && g_rccl.all_reduce) ? 0 : 2;
  });
  return g_rccl;
}
constexpr int kNcclSum = 0, kNcclFloat16 = 6, kNcclFloat32 = 7, kNcclBfloat16 = 9;   // rccl.h ncclRedOp_t / ncclDataType_t
}  // namespace

// 128 bytes that rank 0 creates and every rank of the job must pass to msmd_comm_init (exchange them over any side channel:
// the Python host uses the torch.distributed store it already has for the rendezvous).  Returns 0, or 1000 + n when librccl
// could not be loaded / resolved, or the ncclResult_t.
extern "C" int msmd_comm_unique_id(void* id_out) {
  if (!id_out) return 1;
  const Rccl& r = rccl();
  if (r.status) return 1000 + r.status;
  return r.get_unique_id((UniqueId*)id_out);
}

// Collective: every rank of the job calls it with the same id, its own rank, the current HIP device being the GPU it owns.
extern "C" int msmd_comm_init(void** comm_out, int world, int rank, const void* id) {
  if (!comm_out || !id || world < 1 || rank < 0 || rank >= world) return 1;
  const Rccl& r = rccl();
  if (r.status) return 1000 + r.status;
  UniqueId u;
  std::memcpy(&u, id, sizeof(u));
  return r.comm_init_rank(comm_out, world, u, rank);
}

// ncclGetVersion of the library in use (major * 10000 + minor * 100 + patch), or -(1000 + n) when it could not be resolved.
extern "C" int msmd_comm_version(void) {
  const Rccl& r = rccl();
  return r.status ? -(1000 + r.status) : r.version;
}

extern "C" int msmd_comm_destroy(void* comm) {
  if (!comm) return 1;
  const Rccl& r = rccl();
  if (r.status) return 1000 + r.status;
  return r.comm_destroy(comm);
}

// buf[0 .. n) <- sum over ranks, in place, enqueued on `stream` (asynchronous with the host; no allocation, no host sync):
// one gradient bucket of the flat arena (fp32), or its bf16 / fp16 staging copy.  dtype: MSMD_F32 | MSMD_BF16 | MSMD_F16.
extern "C" int msmd_allreduce_bucket(void* comm, void* buf, long n, int dtype, msmd_stream_t stream) {
  if (!comm || !buf || n <= 0) return 1;
  const Rccl& r = rccl();
  if (r.status) return 1000 + r.status;
  int dt;
  if (dtype == MSMD_F32) dt = kNcclFloat32;
  else if (dtype == MSMD_BF16) dt = kNcclBfloat16;
  else if (dtype == MSMD_F16) dt = kNcclFloat16;
  else return 1;
  return r.all_reduce(buf, buf, (size_t)n, dt, kNcclSum, comm, (hipStream_t)stream);
}
