// loopback_rccl.cpp -- TEST INFRASTRUCTURE: a loop-back stand-in for the few RCCL entry points uc_group.cpp calls, so that
// the group logic at world > 1 (partition offsets, in-place all-gather, the ragged broadcast path, several local devices in
// one ncclGroupStart/End, the write-after-gather guard) can be REHEARSED ON ONE GPU.  RCCL itself refuses two ranks on one
// device, and this build has no multi-GPU box; the real library is what every shipped path loads (librccl.so.1) -- this
// one is only ever loaded when a test sets UC_TUNING=1 UC_RCCL_LIB=<this file's .so>.  "Ranks" may share a device.  Never a
// measurement.  Two forms:
//   one process, several ranks (ncclCommInitAll)       device-to-device copies between the ranks' buffers, stream-ordered
//   one process per rank (ncclCommInitRank, world > 1)  the slices travel through a POSIX shared-memory segment named by
//                                                       the unique id; the calls are SYNCHRONOUS (stream synchronise, copy
//                                                       out, barrier, copy in, barrier) -- good enough to rehearse the
//                                                       launcher's plumbing, nothing like the real transport
//
// Semantics reproduced: collectives are enqueued on the stream each rank names and complete in stream order; a rank's
// contribution is read when THAT rank's stream reaches the call (an event per rank and collective), so a rank that is late
// delays the others exactly as a real collective would.  Operations are collected between ncclGroupStart / ncclGroupEnd
// and issued at ncclGroupEnd, as RCCL does for one thread driving several devices.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <vector>

namespace {

// shared segment of the one-process-per-rank form: a sense-reversing barrier, then the payload
struct Shared {
  std::atomic<unsigned> arrived, generation;
  char pad[56];
  char data[1];
};
constexpr size_t kPayload = 64u << 20;  // bytes a collective may move in all

struct Comm {
  int rank, world, device;
  std::vector<Comm*>* peers;  // one process: all communicators of the clique, by rank; nullptr in the per-process form
  Shared* shm = nullptr;
  char shm_name[40] = {0};
};

void barrier(Comm* c) {
  const unsigned gen = c->shm->generation.load();
  if (c->shm->arrived.fetch_add(1) + 1 == (unsigned)c->world) {
    c->shm->arrived.store(0);
    c->shm->generation.fetch_add(1);
  } else {
    while (c->shm->generation.load() == gen) usleep(50);
  }
}

struct Op {
  int kind;  // 0 all-gather, 1 broadcast
  Comm* comm;
  const void* send;
  void* recv;
  size_t bytes;
  int root;
  hipStream_t stream;
};

thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

ncclResult_t issue(const std::vector<Op*>& by_rank) {
  const int world = (int)by_rank.size();
  // "rank r has reached the collective": an event on its stream, behind everything it enqueued before
  std::vector<hipEvent_t> ready((size_t)world);
  for (int r = 0; r < world; r++) {
    if (hipSetDevice(by_rank[(size_t)r]->comm->device) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventCreateWithFlags(&ready[(size_t)r], hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
    if (hipEventRecord(ready[(size_t)r], by_rank[(size_t)r]->stream) != hipSuccess) return ncclUnhandledCudaError;
  }
  for (int r = 0; r < world; r++) {
    Op& o = *by_rank[(size_t)r];
    if (hipSetDevice(o.comm->device) != hipSuccess) return ncclUnhandledCudaError;
    if (o.kind == 0) {
      for (int q = 0; q < world; q++) {
        if (hipStreamWaitEvent(o.stream, ready[(size_t)q], 0) != hipSuccess) return ncclUnhandledCudaError;
        char* dst = (char*)o.recv + (size_t)q * o.bytes;
        const void* src = by_rank[(size_t)q]->send;
        if (dst != src && hipMemcpyAsync(dst, src, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess)
          return ncclUnhandledCudaError;
      }
    } else {
      const Op& root = *by_rank[(size_t)o.root];
      if (hipStreamWaitEvent(o.stream, ready[(size_t)o.root], 0) != hipSuccess) return ncclUnhandledCudaError;
      if (o.recv != root.send && hipMemcpyAsync(o.recv, root.send, o.bytes, hipMemcpyDeviceToDevice, o.stream) != hipSuccess)
        return ncclUnhandledCudaError;
    }
  }
  for (hipEvent_t e : ready) (void)hipEventDestroy(e);  // (destruction is deferred until the recorded work is done)
  return ncclSuccess;
}

// one process per rank: synchronous, through the shared segment
ncclResult_t issue_shared(Op& o) {
  Comm* c = o.comm;
  if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess) return ncclUnhandledCudaError;
  if (o.kind == 0) {
    if (o.bytes * (size_t)c->world > kPayload) return ncclInvalidArgument;
    if (hipMemcpy(c->shm->data + (size_t)c->rank * o.bytes, o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    if (hipMemcpy(o.recv, c->shm->data, o.bytes * (size_t)c->world, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  } else {
    if (o.bytes > kPayload) return ncclInvalidArgument;
    if (c->rank == o.root && hipMemcpy(c->shm->data, o.send, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    barrier(c);
    if (c->rank != o.root && hipMemcpy(o.recv, c->shm->data, o.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  }
  barrier(c);  // the payload area is free again
  return ncclSuccess;
}

ncclResult_t flush() {
  // the k-th operation of every rank belongs to the k-th collective (each rank issues its operations in the same order)
  if (g_ops.empty()) return ncclSuccess;
  if (g_ops[0].comm->shm) {
    ncclResult_t rc = ncclSuccess;
    for (Op& o : g_ops)
      if (rc == ncclSuccess) rc = issue_shared(o);
    g_ops.clear();
    return rc;
  }
  const int world = g_ops[0].comm->world;
  std::vector<std::vector<Op*>> q((size_t)world);
  for (Op& o : g_ops) {
    if (o.comm->peers != g_ops[0].comm->peers) return ncclInvalidUsage;
    q[(size_t)o.comm->rank].push_back(&o);
  }
  for (int r = 1; r < world; r++)
    if (q[(size_t)r].size() != q[0].size()) return ncclInvalidUsage;
  for (size_t k = 0; k < q[0].size(); k++) {
    std::vector<Op*> by_rank((size_t)world);
    for (int r = 0; r < world; r++) {
      by_rank[(size_t)r] = q[(size_t)r][k];
      if (by_rank[(size_t)r]->kind != by_rank[0]->kind || by_rank[(size_t)r]->root != by_rank[0]->root) return ncclInvalidUsage;
    }
    const ncclResult_t rc = issue(by_rank);
    if (rc != ncclSuccess) return rc;
  }
  g_ops.clear();
  return ncclSuccess;
}

ncclResult_t submit(const Op& o) {
  g_ops.push_back(o);
  if (g_depth == 0) return (o.comm->world == 1 || o.comm->shm) ? flush() : ncclInvalidUsage;  // several ranks from one thread need a group
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetVersion(int* v) { if (v) *v = 0; return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "loop-back RCCL stand-in: error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 0x5a, sizeof(*id));
  FILE* f = fopen("/dev/urandom", "rb");
  if (f) {
    (void)!fread(id->internal, 1, 16, f);
    fclose(f);
  }
  return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
  auto* peers = new std::vector<Comm*>((size_t)ndev);
  for (int r = 0; r < ndev; r++) {
    Comm* c = new Comm{r, ndev, devlist ? devlist[r] : r, peers};
    (*peers)[(size_t)r] = c;
    comms[r] = reinterpret_cast<ncclComm_t>(c);
  }
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (nranks == 1) {
    auto* peers = new std::vector<Comm*>(1);
    Comm* c = new Comm{0, 1, dev, peers};
    (*peers)[0] = c;
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
  }
  Comm* c = new Comm{rank, nranks, dev, nullptr};
  snprintf(c->shm_name, sizeof(c->shm_name), "/uc_lb_%02x%02x%02x%02x%02x%02x%02x%02x", (unsigned char)id.internal[0],
           (unsigned char)id.internal[1], (unsigned char)id.internal[2], (unsigned char)id.internal[3],
           (unsigned char)id.internal[4], (unsigned char)id.internal[5], (unsigned char)id.internal[6], (unsigned char)id.internal[7]);
  const int fd = shm_open(c->shm_name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)(sizeof(Shared) + kPayload)) != 0) { delete c; return ncclSystemError; }
  void* m = mmap(nullptr, sizeof(Shared) + kPayload, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) { delete c; return ncclSystemError; }
  c->shm = static_cast<Shared*>(m);  // (a fresh segment is all zeros: the barrier starts at generation 0)
  barrier(c);                        // everybody has mapped it
  if (rank == 0) shm_unlink(c->shm_name);
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (c->shm) {
    munmap(c->shm, sizeof(Shared) + kPayload);
    delete c;
    return ncclSuccess;
  }
  (*c->peers)[(size_t)c->rank] = nullptr;
  bool last = true;
  for (Comm* p : *c->peers) last = last && p == nullptr;
  if (last) delete c->peers;
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth == 0) return flush();
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm,
                           hipStream_t stream) {
  return submit(Op{0, reinterpret_cast<Comm*>(comm), sendbuff, recvbuff, sendcount * type_bytes(dt), 0, stream});
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t dt, int root, ncclComm_t comm,
                           hipStream_t stream) {
  return submit(Op{1, reinterpret_cast<Comm*>(comm), sendbuff, recvbuff, count * type_bytes(dt), root, stream});
}

}  // extern "C"
