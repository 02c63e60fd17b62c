// Gradient exchange behind the C boundary: an RCCL communicator bootstrapped from an ncclUniqueId that the host side
// broadcasts (one process per GPU), in-place all-reduce on the communicator's own HIP stream, ordered against the
// caller's streams by events only - no call here blocks the host except js2t_comm_wait(comm, NULL) and the bootstrap.
// This is what torch's DistributedDataParallel does for the reference (prediction.py:508-515, helpers_for_ddp.py:17-38,
// 157-174): bucketed all-reduce(average) of the gradients over NCCL; here over RCCL / xGMI.
//
// RCCL is resolved at first use (dlopen): a build without multi-GPU needs has no load-time dependency on it, and a
// process in which torch has already loaded its RCCL shares that copy.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "../../include/joeys2t_hip.h"

void js2t_set_error(const char* fmt, ...);

namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;

void load_rccl() {
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (g_rccl.handle) break;
  }
  if (!g_rccl.handle) return;
  void* h = g_rccl.handle;
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(h, "ncclGetErrorString");
  g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.AllReduce && g_rccl.CommDestroy && g_rccl.GetErrorString;
}

bool rccl_ready() {
  std::call_once(g_once, load_rccl);
  if (!g_rccl.ok) js2t_set_error("js2t_comm: librccl could not be loaded (%s)", g_rccl.handle ? "symbols missing" : dlerror());
  return g_rccl.ok;
}

struct Comm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;  // every collective of this communicator runs here, in call order
  hipEvent_t ready = nullptr;    // "the producer stream has written the buffer"
  hipEvent_t done = nullptr;     // "everything enqueued on `stream` so far"
  int world = 0, rank = 0, device = 0;
};

#define HIP_OK(call, what)                                                   \
  do {                                                                       \
    hipError_t e__ = (call);                                                 \
    if (e__ != hipSuccess) {                                                 \
      js2t_set_error("js2t_comm: %s: %s", what, hipGetErrorString(e__));     \
      return JS2T_ERR_LAUNCH;                                                \
    }                                                                        \
  } while (0)
#define NCCL_OK(call, what)                                                  \
  do {                                                                       \
    ncclResult_t r__ = (call);                                               \
    if (r__ != ncclSuccess) {                                                \
      js2t_set_error("js2t_comm: %s: %s", what, g_rccl.GetErrorString(r__)); \
      return JS2T_ERR_LAUNCH;                                                \
    }                                                                        \
  } while (0)

}  // namespace

extern "C" int64_t js2t_comm_unique_id_bytes(void) { return NCCL_UNIQUE_ID_BYTES; }

// The communicator's stream and events live on c->device: the per-call entry points run with that device current, whatever the
// caller's is, and put the caller's device back.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) ok = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

extern "C" int js2t_comm_unique_id(void* out, int64_t nbytes) {
  if (!out || nbytes < NCCL_UNIQUE_ID_BYTES) {
    js2t_set_error("js2t_comm_unique_id: need a buffer of %d bytes", NCCL_UNIQUE_ID_BYTES);
    return JS2T_ERR_INVALID;
  }
  if (!rccl_ready()) return JS2T_ERR_UNSUPPORTED;
  ncclUniqueId id;
  NCCL_OK(g_rccl.GetUniqueId(&id), "ncclGetUniqueId");
  memcpy(out, id.internal, NCCL_UNIQUE_ID_BYTES);
  return JS2T_OK;
}

extern "C" int js2t_comm_init(void** comm_out, const void* unique_id, int64_t id_bytes, int32_t world, int32_t rank, int32_t device) {
  if (!comm_out || !unique_id || id_bytes != NCCL_UNIQUE_ID_BYTES || world < 1 || rank < 0 || rank >= world || device < 0) {
    js2t_set_error("js2t_comm_init: bad arguments (id of %d bytes, 0 <= rank < world, device >= 0)", NCCL_UNIQUE_ID_BYTES);
    return JS2T_ERR_INVALID;
  }
  if (!rccl_ready()) return JS2T_ERR_UNSUPPORTED;
  int prev = 0;
  HIP_OK(hipGetDevice(&prev), "hipGetDevice");
  HIP_OK(hipSetDevice(device), "hipSetDevice");
  Comm* c = new Comm();  // (after the calls that may return early)
  c->world = world, c->rank = rank, c->device = device;
  ncclUniqueId id;
  memcpy(id.internal, unique_id, NCCL_UNIQUE_ID_BYTES);
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);  // blocks until every rank has arrived
  hipError_t e = hipSuccess;
  if (r == ncclSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (r == ncclSuccess && e == hipSuccess) e = hipEventCreateWithFlags(&c->ready, hipEventDisableTiming);
  if (r == ncclSuccess && e == hipSuccess) e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
  if (r == ncclSuccess && e == hipSuccess) e = hipEventRecord(c->done, c->stream);  // a wait before any collective returns at once
  (void)hipSetDevice(prev);
  if (r != ncclSuccess || e != hipSuccess) {
    js2t_set_error("js2t_comm_init: %s", r != ncclSuccess ? g_rccl.GetErrorString(r) : hipGetErrorString(e));
    if (c->comm) g_rccl.CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    delete c;
    return JS2T_ERR_LAUNCH;
  }
  *comm_out = c;
  return JS2T_OK;
}

extern "C" int js2t_comm_allreduce_async(void* comm, void* buf, int64_t count, int32_t dtype, int32_t average, js2t_stream producer) {
  Comm* c = (Comm*)comm;
  if (!c || !buf || count <= 0 || (dtype != JS2T_F32 && dtype != JS2T_BF16)) {
    js2t_set_error("js2t_comm_allreduce_async: bad arguments (f32 or bf16 buffer, count > 0)");
    return JS2T_ERR_INVALID;
  }
  DeviceGuard guard(c->device);
  if (!guard.ok) {
    js2t_set_error("js2t_comm_allreduce_async: cannot select device %d", c->device);
    return JS2T_ERR_LAUNCH;
  }
  // the collective starts when the producer stream has got as far as this call, and not before
  HIP_OK(hipEventRecord(c->ready, (hipStream_t)producer), "hipEventRecord");
  HIP_OK(hipStreamWaitEvent(c->stream, c->ready, 0), "hipStreamWaitEvent");
  NCCL_OK(g_rccl.AllReduce(buf, buf, (size_t)count, dtype == JS2T_F32 ? ncclFloat32 : ncclBfloat16, average ? ncclAvg : ncclSum, c->comm,
                           c->stream),
          "ncclAllReduce");
  HIP_OK(hipEventRecord(c->done, c->stream), "hipEventRecord");
  return JS2T_OK;
}

extern "C" int js2t_comm_wait(void* comm, js2t_stream consumer, int32_t host) {
  Comm* c = (Comm*)comm;
  if (!c) {
    js2t_set_error("js2t_comm_wait: null communicator");
    return JS2T_ERR_INVALID;
  }
  DeviceGuard guard(c->device);
  if (!guard.ok) {
    js2t_set_error("js2t_comm_wait: cannot select device %d", c->device);
    return JS2T_ERR_LAUNCH;
  }
  if (host) {
    HIP_OK(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
  } else {
    HIP_OK(hipStreamWaitEvent((hipStream_t)consumer, c->done, 0), "hipStreamWaitEvent");
  }
  return JS2T_OK;
}

extern "C" js2t_stream js2t_comm_stream(void* comm) { return comm ? (js2t_stream)((Comm*)comm)->stream : nullptr; }

extern "C" int js2t_comm_destroy(void* comm) {
  Comm* c = (Comm*)comm;
  if (!c) return JS2T_OK;
  DeviceGuard guard(c->device);
  (void)hipStreamSynchronize(c->stream);
  ncclResult_t r = g_rccl.CommDestroy(c->comm);
  (void)hipEventDestroy(c->ready);
  (void)hipEventDestroy(c->done);
  (void)hipStreamDestroy(c->stream);
  delete c;
  if (r != ncclSuccess) {
    js2t_set_error("js2t_comm_destroy: %s", g_rccl.GetErrorString(r));
    return JS2T_ERR_LAUNCH;
  }
  return JS2T_OK;
}
