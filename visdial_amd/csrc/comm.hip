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
  NcclResult (*GetVersion)(int*) = nullptr;      // optional
};

struct CommState {
  Rccl api;
  NcclComm comm = nullptr;
  int rank = 0, world = 1, device = -1;
  hipStream_t stream = nullptr;      // library-owned communication stream
  hipEvent_t ev_tail = nullptr;      // main stream -> comm stream: "the whole backward has been enqueued up to here"
  hipEvent_t ev_done = nullptr;      // comm stream -> main stream: "both buckets are reduced"
  hipEvent_t ev_b1 = nullptr;        // comm stream: bucket 1 reduced        } timed: vd_comm_overlap_ms = how long before the end of the
  hipEvent_t ev_tail_t = nullptr;    // main stream: the backward has ended   } backward the encoder bucket was already summed
  bool timed = false;                // both recorded by the last call
  // the last vd_model_allreduce_grads, for vd_comm_stats
  int64_t calls = 0, bucket1 = 0, bucket2 = 0;
  int overlapped = 0;
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
  *(void**)(&a.GetVersion) = dlsym(lib, "ncclGetVersion");
  g.api = a;
  return VD_OK;
}

// stream + events of this process (not the communicator)
void release_local() {
  if (g.ev_tail) (void)hipEventDestroy(g.ev_tail);
  if (g.ev_done) (void)hipEventDestroy(g.ev_done);
  if (g.ev_b1) (void)hipEventDestroy(g.ev_b1);
  if (g.ev_tail_t) (void)hipEventDestroy(g.ev_tail_t);
  g.ev_b1 = g.ev_tail_t = nullptr;
  g.timed = false;
  if (g.stream) (void)hipStreamDestroy(g.stream);
  g.ev_tail = g.ev_done = nullptr;
  g.stream = nullptr;
  g.comm = nullptr;
  g.world = 1;
  g.rank = 0;
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
  // the stream and events first: nothing below the collective CommInitRank can fail and leave a half-built communicator
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  hipError_t e = hipStreamCreateWithPriority(&g.stream, hipStreamNonBlocking, greatest);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&g.ev_tail, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&g.ev_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&g.ev_b1);
  if (e == hipSuccess) e = hipEventCreate(&g.ev_tail_t);
  NcclResult r = 0;
  if (e == hipSuccess) r = g.api.CommInitRank(&g.comm, world, id, rank);
  if (e != hipSuccess || r != 0) {
    if (e != hipSuccess) vd_set_error("vd_comm_init: %s", hipGetErrorString(e));
    else vd_set_error("vd_comm_init: ncclCommInitRank(rank %d of %d) -> %s", rank, world, g.api.GetErrorString(r));
    release_local();
    return VD_ERR_HIP;
  }
  g.rank = rank;
  g.world = world;
  g.calls = 0;
  return VD_OK;
}

// no collective, no device: can this process load RCCL at all?  A host exchanges the answer over its own channel BEFORE any
// rank enters vd_comm_unique_id / vd_comm_init, so that a rank without a usable librccl cannot strand its peers inside
// the collective ncclCommInitRank.  version = NCCL_VERSION_CODE of the loaded library (0 if it does not say).
int vd_comm_available(int* version) {
  if (version) *version = 0;
  int rc = load_rccl();
  if (rc != VD_OK) return rc;
  if (version && g.api.GetVersion) (void)g.api.GetVersion(version);
  return VD_OK;
}

// the last vd_model_allreduce_grads of this process: floats in bucket 1 (encoder tensors, reduced underneath the decoder's
// backward) and bucket 2 (embedding + decoder), whether bucket 1 was issued early (the model had recorded "encoder
// gradients final" on its side stream), and the number of calls since vd_comm_init
int vd_comm_stats(int64_t* bucket1_floats, int64_t* bucket2_floats, int* overlapped, int64_t* calls) {
  if (bucket1_floats) *bucket1_floats = g.bucket1;
  if (bucket2_floats) *bucket2_floats = g.bucket2;
  if (overlapped) *overlapped = g.overlapped;
  if (calls) *calls = g.calls;
  return VD_OK;
}

// The last vd_model_allreduce_grads: milliseconds between "bucket 1 (the encoder's tensors) is summed over the ranks" on the communication
// stream and "the backward pass has ended" on the main stream.  POSITIVE = the exchange of bucket 1 was hidden under the decoder's backward
// with that much to spare; negative = it finished that long AFTER the backward (exposed).  Waits for both events.  VD_ERR_STATE when the
// last call did not split the gradient (no encoder-final event, or no call yet).
int vd_comm_overlap_ms(float* lead_ms) {
  VD_CHECK_ARG(lead_ms, "vd_comm_overlap_ms: null");
  *lead_ms = 0.f;
  if (!g.comm || !g.timed) {
    vd_set_error("vd_comm_overlap_ms: the last all-reduce was not split into an early and a late bucket");
    return VD_ERR_STATE;
  }
  VD_HIP(hipEventSynchronize(g.ev_b1));
  VD_HIP(hipEventSynchronize(g.ev_tail_t));
  VD_HIP(hipEventElapsedTime(lead_ms, g.ev_b1, g.ev_tail_t));
  return VD_OK;
}

int vd_comm_info(int* rank, int* world) {
  if (rank) *rank = g.comm ? g.rank : 0;
  if (world) *world = g.comm ? g.world : 0;      // 0 = no communicator
  return VD_OK;
}

int vd_comm_destroy(void) {
  if (!g.comm) return VD_OK;
  if (g.stream) (void)hipStreamSynchronize(g.stream);
  NcclResult r = g.api.CommDestroy(g.comm);
  release_local();                       // whatever RCCL said, this process no longer has a communicator
  if (r != 0) {
    vd_set_error("vd_comm_destroy: ncclCommDestroy -> %s", g.api.GetErrorString(r));
    return VD_ERR_HIP;
  }
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
  VdRange r("vd_model_allreduce_grads");
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
    VD_HIP(hipEventRecord(g.ev_b1, g.stream));
    VD_HIP(hipEventRecord(g.ev_tail_t, m->s_main));
  }
  g.timed = split;
  VD_HIP(hipEventRecord(g.ev_tail, m->s_main));
  VD_HIP(hipStreamWaitEvent(g.stream, g.ev_tail, 0));
  if (split) {
    VD_NCCL(g.api.GroupStart());
    NcclResult r = 0;                     // a failure inside the group still closes it
    if (lo > 0) r = g.api.AllReduce(m->G, m->G, (size_t)lo, kNcclFloat32, kNcclSum, g.comm, g.stream);
    if (r == 0 && m->numel > hi)
      r = g.api.AllReduce(m->G + hi, m->G + hi, (size_t)(m->numel - hi), kNcclFloat32, kNcclSum, g.comm, g.stream);
    NcclResult rend = g.api.GroupEnd();
    if (r != 0 || rend != 0) {
      vd_set_error("vd_model_allreduce_grads: bucket 2 -> %s", g.api.GetErrorString(r != 0 ? r : rend));
      return VD_ERR_HIP;
    }
  } else {
    VD_NCCL(g.api.AllReduce(m->G, m->G, (size_t)m->numel, kNcclFloat32, kNcclSum, g.comm, g.stream));
  }
  VD_HIP(hipEventRecord(g.ev_done, g.stream));
  VD_HIP(hipStreamWaitEvent(m->s_main, g.ev_done, 0));
  g.calls++;
  g.overlapped = split ? 1 : 0;
  g.bucket1 = split ? hi - lo : 0;
  g.bucket2 = split ? m->numel - (hi - lo) : m->numel;
  return VD_OK;
}

}  // extern "C"
