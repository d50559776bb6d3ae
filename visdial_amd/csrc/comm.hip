// Data-parallel gradient exchange behind the C ABI (new relative to the single-GPU reference,
// train.lua:19): one RCCL communicator per process (= per GPU), the all-reduce of the flat gradient `wrapperdW` in two
// buckets, issued by the library on its own communication stream.  A LuaJIT host gets data parallelism with three
// extra calls; torch.distributed is not involved.
//
// RCCL is resolved at run time (dlopen of librccl.so.1, which is the nccl API on ROCm): the library has no link-time
// dependency on it, a single-GPU host never loads it, and inside a process that already carries a librccl (PyTorch
// bundles one) the loader hands back that same object instead of a second copy.
#include <dlfcn.h>

#include "common.h"
#include "rt_core.h"
#include "../../include/visdial_hip.h"

namespace {

// the five nccl types / constants this file needs (rccl.h: NCCL_UNIQUE_ID_BYTES = 128, ncclFloat32 = 7, ncclSum = 0)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
typedef int NcclResult;
enum { kNcclFloat32 = 7, kNcclSum = 0 };

struct Rccl {
  void* lib = nullptr;
  NcclResult (*GetUniqueId)(NcclUniqueId*) = nullptr;
  NcclResult (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  NcclResult (*CommDestroy)(NcclComm) = nullptr;
  NcclResult (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  NcclResult (*GroupStart)() = nullptr;
  NcclResult (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(NcclResult) = nullptr;
};

struct CommState {
  Rccl api;
  NcclComm comm = nullptr;
  int rank = 0, world = 1, device = -1;
  hipStream_t stream = nullptr;      // library-owned communication stream
  hipEvent_t ev_tail = nullptr;      // main stream -> comm stream: "the whole backward has been enqueued up to here"
  hipEvent_t ev_done = nullptr;      // comm stream -> main stream: "both buckets are reduced"
};
CommState g;

int load_rccl() {
  if (g.api.lib) return VD_OK;
  const char* override_path = getenv("VD_RCCL_LIB");
  const char* names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
  void* lib = nullptr;
  for (int i = 0; i < 5 && !lib; ++i)
    if (names[i] && names[i][0]) lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    vd_set_error("vd_comm: cannot load librccl.so.1 (%s); set VD_RCCL_LIB to its path", dlerror());
    return VD_ERR_STATE;
  }
  Rccl a;
  a.lib = lib;
#define VD_SYM(field, name)                                              \
  *(void**)(&a.field) = dlsym(lib, name);                                \
  if (!a.field) {                                                        \
    vd_set_error("vd_comm: librccl has no symbol %s", name);             \
    dlclose(lib);                                                        \
    return VD_ERR_STATE;                                                 \
  }
  VD_SYM(GetUniqueId, "ncclGetUniqueId")
  VD_SYM(CommInitRank, "ncclCommInitRank")
  VD_SYM(CommDestroy, "ncclCommDestroy")
  VD_SYM(AllReduce, "ncclAllReduce")
  VD_SYM(GroupStart, "ncclGroupStart")
  VD_SYM(GroupEnd, "ncclGroupEnd")
  VD_SYM(GetErrorString, "ncclGetErrorString")
#undef VD_SYM
  g.api = a;
  return VD_OK;
}

}  // namespace

#define VD_NCCL(expr)                                                                              \
  do {                                                                                             \
    NcclResult r__ = (expr);                                                                       \
    if (r__ != 0) {                                                                                \
      vd_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, g.api.GetErrorString(r__));       \
      return VD_ERR_HIP;                                                                           \
    }                                                                                              \
  } while (0)

extern "C" {

// rank 0: create the 128-byte rendezvous token (ncclGetUniqueId); the host hands it to every rank by whatever channel it
// has (a file, a socket, torch.distributed's store, MPI): the library does no networking of its own
int vd_comm_unique_id(void* out128) {
  VD_CHECK_ARG(out128, "vd_comm_unique_id: null buffer");
  int rc = load_rccl();
  if (rc != VD_OK) return rc;
  NcclUniqueId id;
  VD_NCCL(g.api.GetUniqueId(&id));
  memcpy(out128, id.internal, sizeof(id.internal));
  return VD_OK;
}

// every rank, after vd_set_device: join the communicator (collective call: returns when all `world` ranks have joined)
int vd_comm_init(int rank, int world, const void* id128) {
  VD_CHECK_ARG(id128 && world >= 1 && rank >= 0 && rank < world, "vd_comm_init: bad rank %d / world %d", rank, world);
  VD_CHECK_ARG(!g.comm, "vd_comm_init: a communicator already exists (vd_comm_destroy first)");
  int rc = load_rccl();
  if (rc != VD_OK) return rc;
  NcclUniqueId id;
  memcpy(id.internal, id128, sizeof(id.internal));
  VD_HIP(hipGetDevice(&g.device));
  VD_NCCL(g.api.CommInitRank(&g.comm, world, id, rank));
  g.rank = rank;
  g.world = world;
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  VD_HIP(hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, greatest));
  VD_HIP(hipEventCreateWithFlags(&g.ev_tail, hipEventDisableTiming));
  VD_HIP(hipEventCreateWithFlags(&g.ev_done, hipEventDisableTiming));
  return VD_OK;
}

int vd_comm_info(int* rank, int* world) {
  if (rank) *rank = g.comm ? g.rank : 0;
  if (world) *world = g.comm ? g.world : 0;      // 0 = no communicator
  return VD_OK;
}

int vd_comm_destroy(void) {
  if (!g.comm) return VD_OK;
  (void)hipStreamSynchronize(g.stream);
  VD_NCCL(g.api.CommDestroy(g.comm));
  g.comm = nullptr;
  (void)hipEventDestroy(g.ev_tail);
  (void)hipEventDestroy(g.ev_done);
  (void)hipStreamDestroy(g.stream);
  g.stream = nullptr;
  g.world = 1;
  g.rank = 0;
  return VD_OK;
}

// Sum wrapperdW of the last vd_model_forward_backward over all ranks, in two buckets (enqueue only):
//   bucket 1 = the encoder's own tensors [lo, hi) -- final when the encoder backward ends on its side stream, under a
//              disc decoder several ms before the step ends: reduced on the communication stream UNDERNEATH the
//              option-LSTM backward / weight-gradient kernels of the main stream;
//   bucket 2 = the shared embedding + the decoder's tensors [0, lo) + [hi, numel): behind the main stream's backward.
// The main stream then waits for both, so the next call -- vd_model_update(m, 1 / world) -- sees the reduced gradient.
// Every rank must call this once per step, in the same order.
int vd_model_allreduce_grads(vd_model* m) {
  VD_CHECK_ARG(m, "vd_model_allreduce_grads: null model");
  VD_CHECK_ARG(g.comm, "vd_model_allreduce_grads: no communicator (vd_comm_init first)");
  VD_CHECK_ARG(m->cur >= 0, "vd_model_allreduce_grads: no backward pass has been enqueued");
  int dev = -1;
  VD_HIP(hipGetDevice(&dev));
  VD_CHECK_ARG(dev == g.device, "vd_model_allreduce_grads: communicator lives on device %d, current device is %d", g.device, dev);
  int64_t lo = 0, hi = 0;
  int rc = vd_model_encoder_range(m, &lo, &hi);
  if (rc != VD_OK) return rc;
  const bool split = hi > lo && m->enc_grads_recorded;
  if (split) {
    VD_HIP(hipStreamWaitEvent(g.stream, m->ev_enc_grads, 0));
    VD_NCCL(g.api.AllReduce(m->G + lo, m->G + lo, (size_t)(hi - lo), kNcclFloat32, kNcclSum, g.comm, g.stream));
  }
  VD_HIP(hipEventRecord(g.ev_tail, m->s_main));
  VD_HIP(hipStreamWaitEvent(g.stream, g.ev_tail, 0));
  if (split) {
    VD_NCCL(g.api.GroupStart());
    if (lo > 0) VD_NCCL(g.api.AllReduce(m->G, m->G, (size_t)lo, kNcclFloat32, kNcclSum, g.comm, g.stream));
    if (m->numel > hi)
      VD_NCCL(g.api.AllReduce(m->G + hi, m->G + hi, (size_t)(m->numel - hi), kNcclFloat32, kNcclSum, g.comm, g.stream));
    VD_NCCL(g.api.GroupEnd());
  } else {
    VD_NCCL(g.api.AllReduce(m->G, m->G, (size_t)m->numel, kNcclFloat32, kNcclSum, g.comm, g.stream));
  }
  VD_HIP(hipEventRecord(g.ev_done, g.stream));
  VD_HIP(hipStreamWaitEvent(m->s_main, g.ev_done, 0));
  return VD_OK;
}

}  // extern "C"
