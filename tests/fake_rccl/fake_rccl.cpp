// TEST INFRASTRUCTURE ONLY -- a stand-in *transport* behind the RCCL API, so that the library's native communicator
// (gpflow-slim_amd/csrc/comm_rccl.hip: chunking, roots, slots, stream / event ordering) can be driven by TWO real processes
// on the ONE GPU of the build box, where the real RCCL refuses two ranks on one device.  Collectives are host-blocking and
// staged through POSIX shared memory (device -> host -> shm -> host -> device); nothing here is fast, shipped, or used by the
// product (gps_comm_load(path) is handed this library by tests/test_gpu_comm_native.py only).
// Build with -Wl,-Bsymbolic: the internal calls (ncclBroadcast -> ncclSend ...) must bind to THIS library even when a real
// librccl (PyTorch's) is already in the process' global scope.
// Implements exactly the entry points comm_rccl.hip binds (rccl.h:187,220,260,339,591,611,678,700,722) plus ncclCommAbort.
// Fault injection (tests of the failure paths): FAKE_RCCL_FAIL_RANK=r makes rank r's ncclSend call number
// FAKE_RCCL_FAIL_AFTER + 1 (default 1) fail -- FAKE_RCCL_FAIL_MODE=error (default): it returns ncclSystemError, as a send
// that fails inside an open group does; =stall: it never returns (a rank stuck in the transport).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {
constexpr size_t SLOT_BYTES = 24u << 20;      // one message (a panel of the test sizes is a few MB)
constexpr int MAX_RANKS = 4;
struct Slot { std::atomic<unsigned long long> seq, ack; size_t bytes; };
struct Header { std::atomic<int> attached; Slot slots[MAX_RANKS][MAX_RANKS]; };
struct Comm { int rank, nranks; Header* hdr; char* data; size_t map_bytes; std::string name; unsigned long long sent[MAX_RANKS] = {}, got[MAX_RANKS] = {}; };
struct Pending { int kind; void* buf; size_t bytes; int peer; Comm* c; hipStream_t s; };
thread_local int group_depth = 0;
thread_local std::vector<Pending> queue;

char* slot_data(Comm* c, int from, int to) { return c->data + ((size_t)from * MAX_RANKS + to) * SLOT_BYTES; }

template <class F> bool spin(F f) {
  const auto t0 = std::chrono::steady_clock::now();
  while (!f()) {
    static const int limit_s = getenv("FAKE_RCCL_TIMEOUT_S") ? atoi(getenv("FAKE_RCCL_TIMEOUT_S")) : 60;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(limit_s)) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
  return true;
}

ncclResult_t do_send(Comm* c, const void* dev, size_t bytes, int peer, hipStream_t s) {
  if (bytes > SLOT_BYTES) return ncclInvalidArgument;
  Slot& sl = c->hdr->slots[c->rank][peer];
  if (!spin([&] { return sl.ack.load() == c->sent[peer]; })) return ncclSystemError;       // previous message consumed
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(slot_data(c, c->rank, peer), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  sl.bytes = bytes;
  sl.seq.store(++c->sent[peer]);
  return ncclSuccess;
}
ncclResult_t do_recv(Comm* c, void* dev, size_t bytes, int peer, hipStream_t s) {
  Slot& sl = c->hdr->slots[peer][c->rank];
  if (!spin([&] { return sl.seq.load() == c->got[peer] + 1; })) return ncclSystemError;
  if (sl.bytes != bytes) return ncclInvalidArgument;
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(dev, slot_data(c, peer, c->rank), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  sl.ack.store(++c->got[peer]);
  return ncclSuccess;
}
ncclResult_t flush() {
  // all sends first (slots are buffered per ordered pair), then all receives: no rendezvous deadlock
  ncclResult_t rc = ncclSuccess;
  for (auto& p : queue) if (p.kind == 0 && rc == ncclSuccess) rc = do_send(p.c, p.buf, p.bytes, p.peer, p.s);
  for (auto& p : queue) if (p.kind == 1 && rc == ncclSuccess) rc = do_recv(p.c, p.buf, p.bytes, p.peer, p.s);
  queue.clear();
  return rc;
}
size_t dsize(ncclDataType_t t) { return (t == ncclDouble || t == ncclInt64 || t == ncclUint64) ? 8 : ((t == ncclFloat || t == ncclInt32 || t == ncclUint32) ? 4 : 1); }
}  // namespace

extern "C" {
ncclResult_t ncclGetVersion(int* v) { *v = 29999; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl: error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0, sizeof(*id));
  unsigned long long v = (unsigned long long)getpid() * 1000003ull ^ (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count();
  snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%016llx", v);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (nranks > MAX_RANKS) return ncclInvalidArgument;
  Comm* c = new Comm;
  c->rank = rank; c->nranks = nranks; c->name = id.internal;
  c->map_bytes = sizeof(Header) + (size_t)MAX_RANKS * MAX_RANKS * SLOT_BYTES;
  const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return ncclSystemError;
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->hdr = reinterpret_cast<Header*>(p);                      // (a fresh segment is zero filled: counters start at 0)
  c->data = reinterpret_cast<char*>(p) + sizeof(Header);
  c->hdr->attached.fetch_add(1);
  if (!spin([&] { return c->hdr->attached.load() >= nranks; })) return ncclSystemError;
  *out = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  shm_unlink(c->name.c_str());
  munmap(c->hdr, c->map_bytes);
  delete c;
  return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }
ncclResult_t ncclGroupStart() { ++group_depth; return ncclSuccess; }
ncclResult_t ncclGroupEnd() { if (--group_depth == 0) return flush(); return ncclSuccess; }
static bool inject_fault(Comm* c) {
  static const int fail_rank = getenv("FAKE_RCCL_FAIL_RANK") ? atoi(getenv("FAKE_RCCL_FAIL_RANK")) : -1;
  static const long fail_after = getenv("FAKE_RCCL_FAIL_AFTER") ? atol(getenv("FAKE_RCCL_FAIL_AFTER")) : 0;
  static const bool stall = getenv("FAKE_RCCL_FAIL_MODE") && !strcmp(getenv("FAKE_RCCL_FAIL_MODE"), "stall");
  static std::atomic<long> calls{0};
  if (c->rank != fail_rank || ++calls <= fail_after) return false;
  if (stall) for (;;) std::this_thread::sleep_for(std::chrono::seconds(1));
  return true;
}
ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  if (inject_fault(reinterpret_cast<Comm*>(comm))) return ncclSystemError;
  queue.push_back({0, const_cast<void*>(buf), count * dsize(t), peer, reinterpret_cast<Comm*>(comm), s});
  return group_depth ? ncclSuccess : flush();
}
ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  queue.push_back({1, buf, count * dsize(t), peer, reinterpret_cast<Comm*>(comm), s});
  return group_depth ? ncclSuccess : flush();
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t comm, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  ncclGroupStart();
  if (c->rank == root) { for (int p = 0; p < c->nranks; ++p) if (p != root) ncclSend(send, count, t, p, comm, s); }
  else ncclRecv(recv, count, t, root, comm, s);
  return ncclGroupEnd();
}
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t b = count * dsize(t);
  char* out = reinterpret_cast<char*>(recv);
  if (hipStreamSynchronize(s) != hipSuccess) return ncclUnhandledCudaError;
  if (send != out + (size_t)c->rank * b && hipMemcpy(out + (size_t)c->rank * b, send, b, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
  ncclGroupStart();
  for (int p = 0; p < c->nranks; ++p) if (p != c->rank) { ncclSend(out + (size_t)c->rank * b, count, t, p, comm, s); ncclRecv(out + (size_t)p * b, count, t, p, comm, s); }
  return ncclGroupEnd();
}
ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (t != ncclDouble || op != ncclSum) return ncclInvalidArgument;
  const size_t b = count * 8;
  if (b > SLOT_BYTES) return ncclInvalidArgument;
  std::vector<double> mine(count), acc(count, 0.0), other(count);
  if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(mine.data(), send, b, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
  // every rank posts its operand to every peer, then adds the operands in rank order (the same bits everywhere)
  for (int p = 0; p < c->nranks; ++p) if (p != c->rank) {
    Slot& sl = c->hdr->slots[c->rank][p];
    if (!spin([&] { return sl.ack.load() == c->sent[p]; })) return ncclSystemError;
    memcpy(slot_data(c, c->rank, p), mine.data(), b); sl.bytes = b; sl.seq.store(++c->sent[p]);
  }
  for (int p = 0; p < c->nranks; ++p) {
    const double* src = mine.data();
    if (p != c->rank) {
      Slot& sl = c->hdr->slots[p][c->rank];
      if (!spin([&] { return sl.seq.load() == c->got[p] + 1; })) return ncclSystemError;
      memcpy(other.data(), slot_data(c, p, c->rank), b); sl.ack.store(++c->got[p]);
      src = other.data();
    }
    for (size_t i = 0; i < count; ++i) acc[i] += src[i];
  }
  if (hipMemcpy(recv, acc.data(), b, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}
}
