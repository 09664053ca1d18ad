// Native collectives: RCCL called directly from the library (no PyTorch in the data path).
//
// There is no reference counterpart (GPflow-Slim is single-device); this is the exchange step of SURVEY 8(e):
// "ncclBroadcast (RCCL) per panel, root = owner, on a dedicated stream; on the fully-connected xGMI mesh equivalently
// 7 concurrent peer writes", and the scalar / [M, M] all-reduce of the data-sharded sparse models.
//
// librccl is opened at run time (dlopen: the library itself links libamdhip64 only, and a process that never asks for a
// communicator never loads RCCL).  One communicator per handle, one rank per process (or per host thread), collectives on a
// stream of their own that is ordered against the handle's stream with events, so that an exchange overlaps the kernels the
// schedule issues after it:
//     gps_comm_exchange(h, buf, count, root, mode, slot)   "buf is complete on root" -> complete everywhere; returns at once
//     gps_comm_wait(h, slot)                               the handle's stream waits for that exchange
// mode 0: ncclBroadcast.  mode 1: the root scatters P equal chunks (ncclSend / ncclRecv in one group: all of its xGMI links
// carry 1 / P of the panel at once) and an in-place ncclAllGather completes it -- 2 S / (P b) per panel instead of S / b.
#include "gps_common.hpp"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>       // types and prototypes only; every symbol is bound with dlsym

namespace {
struct RcclApi {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;          // optional: absent from a transport stand-in
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};
RcclApi g_api;
std::mutex g_api_mutex;
std::string g_api_error;

template <class F>
bool bind(void* lib, const char* name, F& fn) {
  fn = reinterpret_cast<F>(dlsym(lib, name));
  if (!fn) g_api_error = std::string("librccl: symbol not found: ") + name;
  return fn != nullptr;
}

int load_api(const char* path) {
  std::lock_guard<std::mutex> lock(g_api_mutex);
  if (g_api.lib) return GPS_OK;
  const char* names[] = {path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names) {
    if (!n || !*n) continue;
    lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (lib) break;
    const char* e = dlerror();                      // (a second dlerror() would return NULL: the first call clears it)
    g_api_error = std::string("dlopen: ") + (e ? e : "?");
  }
  if (!lib) return GPS_ERR_UNSUPPORTED;
  RcclApi a;
  a.lib = lib;
  const bool ok = bind(lib, "ncclGetUniqueId", a.GetUniqueId) && bind(lib, "ncclCommInitRank", a.CommInitRank) &&
                  bind(lib, "ncclCommDestroy", a.CommDestroy) && bind(lib, "ncclGetErrorString", a.GetErrorString) &&
                  bind(lib, "ncclBroadcast", a.Broadcast) && bind(lib, "ncclAllReduce", a.AllReduce) &&
                  bind(lib, "ncclAllGather", a.AllGather) && bind(lib, "ncclSend", a.Send) && bind(lib, "ncclRecv", a.Recv) &&
                  bind(lib, "ncclGroupStart", a.GroupStart) && bind(lib, "ncclGroupEnd", a.GroupEnd) &&
                  bind(lib, "ncclGetVersion", a.GetVersion);
  if (!ok) { dlclose(lib); return GPS_ERR_UNSUPPORTED; }
  a.CommAbort = reinterpret_cast<decltype(a.CommAbort)>(dlsym(lib, "ncclCommAbort"));
  g_api = a;
  return GPS_OK;
}

#define GPS_NCCL(h, call)                                                                          \
  do {                                                                                             \
    ncclResult_t r__ = (call);                                                                     \
    if (r__ != ncclSuccess)                                                                        \
      return gps_fail(h, GPS_ERR_HIP, std::string(#call) + ": " + g_api.GetErrorString(r__));      \
  } while (0)

// A collective that failed on this rank leaves the peers inside theirs: abort the communicator so that they come back with an
// error instead of waiting for this rank's part for ever (ncclCommAbort tears the transport down; the handle then has no
// communicator and every later gps_comm_* call says so).  Returns `code` for `return comm_failed(...)`.
int comm_failed(gps_handle_t h, int code, const std::string& what, ncclResult_t r) {
  std::string msg = what + ": " + (g_api.GetErrorString ? g_api.GetErrorString(r) : "?");
  if (h->comm) {
    ncclComm_t comm = reinterpret_cast<ncclComm_t>(h->comm);
    h->comm = nullptr;
    if (g_api.CommAbort) (void)g_api.CommAbort(comm); else (void)g_api.CommDestroy(comm);
    msg += " (communicator aborted)";
  }
  return gps_fail(h, code, msg);
}

int allreduce_cb(void* ctx, void* dev_ptr, int64_t count) {
  return gps_comm_allreduce(reinterpret_cast<gps_handle_t>(ctx), dev_ptr, count);
}
}  // namespace

extern "C" int gps_comm_load(const char* path) { return load_api(path); }

extern "C" const char* gps_comm_load_error(void) { return g_api_error.c_str(); }

extern "C" int gps_comm_version(int* version) {
  if (!version) return GPS_ERR_ARG;
  if (load_api(nullptr)) return GPS_ERR_UNSUPPORTED;
  return g_api.GetVersion(version) == ncclSuccess ? GPS_OK : GPS_ERR_HIP;
}

extern "C" int gps_comm_unique_id(void* out, int capacity) {
  if (!out || capacity < (int)sizeof(ncclUniqueId)) return GPS_ERR_ARG;
  if (load_api(nullptr)) return GPS_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (g_api.GetUniqueId(&id) != ncclSuccess) return GPS_ERR_HIP;
  memcpy(out, &id, sizeof(id));
  return GPS_OK;
}

extern "C" int gps_comm_init(gps_handle_t h, int rank, int world, const void* unique_id, int id_len) {
  if (!h || !unique_id || id_len < (int)sizeof(ncclUniqueId) || world <= 0 || rank < 0 || rank >= world)
    return gps_fail(h, GPS_ERR_ARG, "gps_comm_init: bad argument");
  if (load_api(nullptr)) return gps_fail(h, GPS_ERR_UNSUPPORTED, "gps_comm_init: librccl could not be loaded: " + g_api_error);
  if (h->comm) return gps_fail(h, GPS_ERR_STATE, "gps_comm_init: the handle already has a communicator");
  GPS_HIP(h, hipSetDevice(h->device));
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  GPS_NCCL(h, g_api.CommInitRank(&comm, world, id, rank));
  h->comm = comm; h->comm_rank = rank; h->comm_world = world;
  GPS_HIP(h, hipStreamCreateWithFlags(&h->comm_stream, hipStreamNonBlocking));
  GPS_HIP(h, hipEventCreateWithFlags(&h->comm_ready, hipEventDisableTiming));
  for (int i = 0; i < 8; ++i) GPS_HIP(h, hipEventCreateWithFlags(&h->comm_done[i], hipEventDisableTiming));
  return GPS_OK;
}

extern "C" int gps_comm_destroy(gps_handle_t h) {
  if (!h) return GPS_ERR_ARG;
  if (!h->comm && !h->comm_stream) return GPS_OK;
  (void)hipSetDevice(h->device);
  if (h->comm && h->comm_stream) (void)hipStreamSynchronize(h->comm_stream);      // (an aborted communicator's stream is not waited for)
  if (h->allreduce == allreduce_cb) { h->allreduce = nullptr; h->allreduce_ctx = nullptr; h->red_buf = nullptr; h->red_cap = 0; }
  if (h->comm) (void)g_api.CommDestroy(reinterpret_cast<ncclComm_t>(h->comm));
  h->comm = nullptr;
  if (h->comm_stream) { (void)hipStreamDestroy(h->comm_stream); h->comm_stream = nullptr; }
  if (h->comm_ready) { (void)hipEventDestroy(h->comm_ready); h->comm_ready = nullptr; }
  for (int i = 0; i < 8; ++i) if (h->comm_done[i]) { (void)hipEventDestroy(h->comm_done[i]); h->comm_done[i] = nullptr; }
  return GPS_OK;
}

// Give up the communicator without waiting for the peers (a rank that has failed outside a collective calls this so that the
// others' pending collectives return with an error; bench.py's watchdog).  Safe without a communicator.
extern "C" int gps_comm_abort(gps_handle_t h) {
  if (!h) return GPS_ERR_ARG;
  if (!h->comm) return GPS_OK;
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(h->comm);
  h->comm = nullptr;
  if (g_api.CommAbort) (void)g_api.CommAbort(comm); else (void)g_api.CommDestroy(comm);
  return GPS_OK;
}

extern "C" int gps_comm_exchange(gps_handle_t h, void* dev_buf, int64_t count, int root, int mode, int slot) {
  if (!h || !h->comm) return gps_fail(h, GPS_ERR_STATE, "gps_comm_exchange: no communicator (gps_comm_init)");
  if (!dev_buf || count <= 0 || root < 0 || root >= h->comm_world || slot < 0 || slot >= 8)
    return gps_fail(h, GPS_ERR_ARG, "gps_comm_exchange: bad argument");
  GPS_HIP(h, hipSetDevice(h->device));
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(h->comm);
  const int P = h->comm_world, me = h->comm_rank;
  double* buf = reinterpret_cast<double*>(dev_buf);
  // everything the handle's stream has done to the buffer so far (the owner's pack; the last reader of this slot) first
  GPS_HIP(h, hipEventRecord(h->comm_ready, h->stream));
  GPS_HIP(h, hipStreamWaitEvent(h->comm_stream, h->comm_ready, 0));
  if (P > 1) {
    const i64 chunk = count / P;
    ncclResult_t r = ncclSuccess;
    const char* at = "";
    if (mode == 1 && chunk > 0) {
      r = g_api.GroupStart(); at = "ncclGroupStart";
      if (r == ncclSuccess) {
        // whatever a send / receive returns, the group is closed again: an open group would swallow every later call
        ncclResult_t rg = ncclSuccess;
        if (me == root) {
          for (int p = 0; p < P && rg == ncclSuccess; ++p)
            if (p != root) { rg = g_api.Send(buf + (i64)p * chunk, (size_t)chunk, ncclDouble, p, comm, h->comm_stream); at = "ncclSend"; }
        } else {
          rg = g_api.Recv(buf + (i64)me * chunk, (size_t)chunk, ncclDouble, root, comm, h->comm_stream); at = "ncclRecv";
        }
        const ncclResult_t re = g_api.GroupEnd();
        if (rg != ncclSuccess) r = rg; else if (re != ncclSuccess) { r = re; at = "ncclGroupEnd"; }
      }
      if (r == ncclSuccess) { r = g_api.AllGather(buf + (i64)me * chunk, buf, (size_t)chunk, ncclDouble, comm, h->comm_stream); at = "ncclAllGather"; }
      if (r == ncclSuccess && chunk * P < count) {       // ragged end of the message
        r = g_api.Broadcast(buf + chunk * P, buf + chunk * P, (size_t)(count - chunk * P), ncclDouble, root, comm, h->comm_stream);
        at = "ncclBroadcast";
      }
    } else {
      r = g_api.Broadcast(buf, buf, (size_t)count, ncclDouble, root, comm, h->comm_stream); at = "ncclBroadcast";
    }
    if (r != ncclSuccess) return comm_failed(h, GPS_ERR_HIP, std::string("gps_comm_exchange: ") + at, r);
  }
  GPS_HIP(h, hipEventRecord(h->comm_done[slot], h->comm_stream));
  return GPS_OK;
}

extern "C" int gps_comm_wait(gps_handle_t h, int slot) {
  if (!h || !h->comm || slot < 0 || slot >= 8) return gps_fail(h, GPS_ERR_ARG, "gps_comm_wait: bad argument");
  GPS_HIP(h, hipStreamWaitEvent(h->stream, h->comm_done[slot], 0));
  return GPS_OK;
}

// in-place sum over ranks of `count` doubles at dev_ptr; returns when the result is visible (blocking)
extern "C" int gps_comm_allreduce(gps_handle_t h, void* dev_ptr, int64_t count) {
  if (!h || !h->comm) return gps_fail(h, GPS_ERR_STATE, "gps_comm_allreduce: no communicator (gps_comm_init)");
  if (!dev_ptr || count <= 0) return gps_fail(h, GPS_ERR_ARG, "gps_comm_allreduce: bad argument");
  GPS_HIP(h, hipSetDevice(h->device));
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(h->comm);
  const ncclResult_t r = g_api.AllReduce(dev_ptr, dev_ptr, (size_t)count, ncclDouble, ncclSum, comm, h->stream);
  if (r != ncclSuccess) return comm_failed(h, GPS_ERR_HIP, "gps_comm_allreduce: ncclAllReduce", r);
  GPS_HIP(h, hipStreamSynchronize(h->stream));
  return GPS_OK;
}

// the data-sharded sparse models reduce through this communicator (gps_set_allreduce with a C callback: no host language
// in the loop); dev_buf: caller-owned device buffer of at least gps_allreduce_doubles(m, r) doubles
extern "C" int gps_comm_install_allreduce(gps_handle_t h, void* dev_buf, int64_t capacity_doubles) {
  if (!h || !h->comm) return gps_fail(h, GPS_ERR_STATE, "gps_comm_install_allreduce: no communicator (gps_comm_init)");
  return gps_set_allreduce(h, allreduce_cb, h, dev_buf, capacity_doubles);
}
