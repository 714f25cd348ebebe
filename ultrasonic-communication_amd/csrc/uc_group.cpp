// uc_group.cpp -- the frame-sharded multi-GPU leg behind the C-ABI (include/uchirp.h: uc_group_*, uc_partition,
// uc_frame_span, uc_stream_span, uc_device_*).
//
// What the north star asks of N GPUs: the reference's C host (receiver/Src/main.c:311-587) drives ONE sample stream; here
// a C host drives a node.  Frames are independent, so rank r decodes a contiguous block of the frame
// index space and the only exchange is the all-gather of the symbol stream, 1 byte per frame -- latency-bound on xGMI
// (1 MiB per GPU per step at the bench's size), issued on a side stream behind an event so that it overlaps the next
// step's kernel.  RCCL is called directly (ncclAllGather / grouped ncclBroadcast), from one thread, for all local devices
// inside one ncclGroupStart/End.  The library is loaded with dlopen when the first group is built.
#include <dlfcn.h>
#include <errno.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <new>
#include <vector>

#include "../../include/uchirp.h"
#include "uc_rx.hpp"

namespace uc {
void set_error(const char* msg);  // uc_api_core.cpp: the thread's uc_last_error() text
}

namespace {

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  uc::set_error(buf);
  return code;
}

int hip_fail(hipError_t e, const char* what) { return fail(-EIO, "%s: %s", what, hipGetErrorString(e)); }

bool tuning_on() {
  const char* t = getenv("UC_TUNING");
  return t && atoi(t) != 0;
}

// ---- RCCL, loaded on first use ---------------------------------------------------------------------------------------
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
Rccl g_rccl;

std::mutex g_rccl_mutex;  // (groups may be built from several threads; the table below is filled once)

int load_rccl() {
  const std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return 0;
  // by soname first: a process that already holds an RCCL (PyTorch bundles one with the same soname) gets THAT one back
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  // Rehearsal hook (read only under UC_TUNING=1, like every other experiment switch): another library with the same entry
  // points.  tests/stubs/loopback_rccl.cpp uses it to run the group logic at world > 1 on ONE GPU (RCCL refuses two ranks
  // on one device).  Never a measurement.
  if (tuning_on())
    if (const char* over = getenv("UC_RCCL_LIB")) {
      h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
      if (!h) return fail(-ENOSYS, "uc_group: UC_RCCL_LIB=%s: %s", over, dlerror());
    }
  for (const char* nm : names) {
    if (h) break;
    h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
  }
  if (!h) return fail(-ENOSYS, "uc_group: librccl.so.1 not found (%s)", dlerror());
  Rccl r;
  r.handle = h;
#define UC_SYM(field, name)                                                                 \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));                            \
  if (!r.field) { dlclose(h); return fail(-ENOSYS, "uc_group: librccl lacks %s", name); }
  UC_SYM(GetVersion, "ncclGetVersion")
  UC_SYM(GetUniqueId, "ncclGetUniqueId")
  UC_SYM(CommInitRank, "ncclCommInitRank")
  UC_SYM(CommInitAll, "ncclCommInitAll")
  UC_SYM(CommDestroy, "ncclCommDestroy")
  UC_SYM(AllGather, "ncclAllGather")
  UC_SYM(Broadcast, "ncclBroadcast")
  UC_SYM(GroupStart, "ncclGroupStart")
  UC_SYM(GroupEnd, "ncclGroupEnd")
  UC_SYM(GetErrorString, "ncclGetErrorString")
#undef UC_SYM
  g_rccl = r;
  return 0;
}

int nccl_fail(ncclResult_t r, const char* what) {
  return fail(-EIO, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}

bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

constexpr int kHazardRing = 8;  // gathers remembered per device for the write-after-gather guard

struct Local {
  int device = 0;
  uc_ctx* ctx = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t compute = nullptr;  // the group's own launch stream (used when the caller names none)
  hipStream_t gather = nullptr;   // the all-gather runs here, behind kernel_done
  hipEvent_t kernel_done = nullptr;
  // write-after-gather guard: the last kHazardRing gathers of this device, (buffer, bytes, event recorded behind it)
  const uint8_t* hz_buf[kHazardRing] = {};
  size_t hz_bytes[kHazardRing] = {};
  hipEvent_t hz_ev[kHazardRing] = {};
  unsigned hz_next = 0;
  // host-pointer calls: the stream (slot 0; the texts of uc_group_receive_streams) and the per-stream counts (slot 1) are
  // gathered here, then copied out
  void* d_stage[2] = {};
  size_t d_stage_cap[2] = {};
};

// a device buffer of at least `bytes` for a host-pointer call (kept, grown on demand)
int stage_buffer(Local& L, int slot, size_t bytes, uint8_t** out) {
  if (L.d_stage_cap[slot] < bytes) {
    if (L.d_stage[slot]) (void)hipFree(L.d_stage[slot]);
    L.d_stage[slot] = nullptr;
    L.d_stage_cap[slot] = 0;
    const hipError_t e = hipMalloc(&L.d_stage[slot], bytes);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(gathered)");
    L.d_stage_cap[slot] = bytes;
  }
  *out = (uint8_t*)L.d_stage[slot];
  return 0;
}

bool overlaps(const uint8_t* a, size_t na, const uint8_t* b, size_t nb) { return a < b + nb && b < a + na; }

// write-after-gather: an earlier gather that still reads or writes [d, d + bytes) must be done before a kernel on `cs`
// overwrites the rank's slice of it (device-side wait, nothing blocks here) ... and the `n_new` ring slots this step will
// recycle: a gather that drops out of the ring must be complete before anything newer runs, or a caller rotating more than
// kHazardRing buffers could overwrite one behind the guard's back
int hazard_wait(Local& L, hipStream_t cs, const uint8_t* d, size_t bytes, unsigned n_new) {
  for (int k = 0; k < kHazardRing; k++) {
    if (!L.hz_buf[k]) continue;
    const bool recycled = (unsigned)(k + kHazardRing - (int)(L.hz_next % kHazardRing)) % kHazardRing < n_new;
    if (recycled || overlaps(L.hz_buf[k], L.hz_bytes[k], d, bytes)) {
      const hipError_t e = hipStreamWaitEvent(cs, L.hz_ev[k], 0);
      if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent(gather -> kernel)");
    }
  }
  return 0;
}

// remember a gather into [d, d + bytes) that was just enqueued on the device's gather stream
int hazard_record(Local& L, const uint8_t* d, size_t bytes) {
  const unsigned k = L.hz_next++ % kHazardRing;
  const hipError_t e = hipEventRecord(L.hz_ev[k], L.gather);
  if (e != hipSuccess) return hip_fail(e, "hipEventRecord(gather)");
  L.hz_buf[k] = d;
  L.hz_bytes[k] = bytes;
  return 0;
}

// The in-place all-gather of n_units_total units of unit_bytes each, block-partitioned over the ranks as uc_partition says:
// ncclAllGather when the shares are even, one broadcast per rank (each slice from its owner) when they are ragged.
// Call between ncclGroupStart and ncclGroupEnd.
ncclResult_t gather_in_place(int world, int rank, uint8_t* d, size_t unit_bytes, size_t n_units_total, ncclComm_t comm,
                             hipStream_t stream) {
  if (n_units_total % (size_t)world == 0) {
    const size_t per = n_units_total / (size_t)world * unit_bytes;
    return g_rccl.AllGather(d + (size_t)rank * per, d, per, ncclUint8, comm, stream);
  }
  ncclResult_t r = ncclSuccess;
  for (int root = 0; root < world && r == ncclSuccess; root++) {
    size_t first = 0, count = 0;
    uc_partition(n_units_total, world, root, &first, &count);
    if (count) r = g_rccl.Broadcast(d + first * unit_bytes, d + first * unit_bytes, count * unit_bytes, ncclUint8, root, comm, stream);
  }
  return r;
}

}  // namespace

struct uc_group {
  int world = 0, first_rank = 0;
  std::vector<Local> loc;
};

extern "C" {

int uc_partition(size_t n_units, int world, int rank, size_t* first, size_t* count) {
  if (world <= 0 || rank < 0 || rank >= world) return fail(-EINVAL, "uc_partition: rank %d of %d", rank, world);
  const size_t q = n_units / (size_t)world, r = n_units % (size_t)world, rk = (size_t)rank;
  if (first) *first = rk * q + (rk < r ? rk : r);
  if (count) *count = q + (rk < r ? 1 : 0);
  return 0;
}

int uc_frame_span(uint32_t n, size_t stride_elems, size_t halo, size_t first_frame, size_t count, size_t* first_elem,
                  size_t* n_elems) {
  if (n == 0) return fail(-EINVAL, "uc_frame_span: n is 0");
  if (stride_elems == 0) stride_elems = n;
  if (first_elem) *first_elem = count ? first_frame * stride_elems : 0;
  if (n_elems) *n_elems = count ? halo + (count - 1) * stride_elems + n : 0;
  return 0;
}

int uc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int uc_device_malloc(int device, size_t bytes, void** out) {
  if (!out) return fail(-EINVAL, "uc_device_malloc: out is NULL");
  *out = nullptr;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  e = hipMalloc(out, bytes ? bytes : 1);
  if (e != hipSuccess) return hip_fail(e, "hipMalloc");
  return 0;
}

int uc_device_free(int device, void* ptr) {
  if (!ptr) return 0;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  e = hipFree(ptr);
  if (e != hipSuccess) return hip_fail(e, "hipFree");
  return 0;
}

int uc_device_copy(void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return 0;
  if (!dst || !src) return fail(-EINVAL, "uc_device_copy: NULL pointer");
  const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyDefault);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy");
  return 0;
}

int uc_group_unique_id(void* id, size_t cap) {
  static_assert(UC_GROUP_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "uc_group id is an ncclUniqueId");
  if (!id || cap < UC_GROUP_ID_BYTES) return fail(-EINVAL, "uc_group_unique_id: need %d bytes", UC_GROUP_ID_BYTES);
  const int rc = load_rccl();
  if (rc) return rc;
  ncclUniqueId u;
  const ncclResult_t r = g_rccl.GetUniqueId(&u);
  if (r != ncclSuccess) return nccl_fail(r, "ncclGetUniqueId");
  memcpy(id, u.internal, UC_GROUP_ID_BYTES);
  return 0;
}

void uc_group_destroy(uc_group* g) {
  if (!g) return;
  for (Local& L : g->loc) {
    if (!L.ctx) continue;  // (never came to life -- a bad ordinal, say: nothing of it to wait for or free)
    (void)hipSetDevice(L.device);
    if (L.compute) (void)hipStreamSynchronize(L.compute);
    if (L.gather) (void)hipStreamSynchronize(L.gather);
  }
  for (Local& L : g->loc) {
    if (!L.ctx) continue;
    (void)hipSetDevice(L.device);
    if (L.comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(L.comm);
    if (L.ctx) uc_destroy(L.ctx);
    for (hipEvent_t ev : L.hz_ev)
      if (ev) (void)hipEventDestroy(ev);
    if (L.kernel_done) (void)hipEventDestroy(L.kernel_done);
    if (L.compute) (void)hipStreamDestroy(L.compute);
    if (L.gather) (void)hipStreamDestroy(L.gather);
    for (void* p : L.d_stage)
      if (p) (void)hipFree(p);
  }
  (void)hipGetLastError();  // (HIP's last-error slot is sticky: leave nothing of the teardown for the next launch to report)
  delete g;
}

// contexts, streams and events of the local devices (the communicators are made by the callers below)
static int group_locals(uc_group* g, const uc_config* cfg, const int32_t* devices, int n) {
  g->loc.resize((size_t)n);
  for (int l = 0; l < n; l++) {
    Local& L = g->loc[(size_t)l];
    L.device = devices[l];
    uc_config c = *cfg;
    c.device = devices[l];
    int rc = uc_create(&c, &L.ctx);
    if (rc) return rc;
    hipError_t e = hipSetDevice(L.device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&L.compute, hipStreamNonBlocking);
    // The gather stream gets the HIGHEST priority the device offers: the decode kernels are persistent and fill every CU, so
    // RCCL's copy kernel is dispatched only as workgroups retire -- with a priority above theirs it is the first thing
    // the dispatcher places when that happens, instead of competing with the next decode launch's workgroups for the slots.
    // (1 MiB per GPU and step: latency-bound on xGMI; what matters is when the kernel STARTS.)
    if (e == hipSuccess) {
      int least = 0, greatest = 0;
      if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
        (void)hipGetLastError();
        least = greatest = 0;
      }
      e = hipStreamCreateWithPriority(&L.gather, hipStreamNonBlocking, greatest);
      if (e != hipSuccess) {  // (no priorities on this device / runtime: a plain stream does the same work)
        (void)hipGetLastError();
        e = hipStreamCreateWithFlags(&L.gather, hipStreamNonBlocking);
      }
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&L.kernel_done, hipEventDisableTiming);
    for (int k = 0; e == hipSuccess && k < kHazardRing; k++) e = hipEventCreateWithFlags(&L.hz_ev[k], hipEventDisableTiming);
    if (e != hipSuccess) return hip_fail(e, "uc_group: stream / event creation");
  }
  return 0;
}

int uc_group_create(const uc_config* cfg, const int32_t* devices, int n_devices, uc_group** out) {
  if (!cfg || !devices || !out) return fail(-EINVAL, "uc_group_create: NULL argument");
  *out = nullptr;
  if (n_devices <= 0 || n_devices > 64) return fail(-EINVAL, "uc_group_create: %d devices", n_devices);
  // (rehearsal on one GPU, UC_TUNING=1 UC_GROUP_SHARE_DEVICES=1 with the loop-back library: several ranks on one device)
  const bool share = tuning_on() && getenv("UC_GROUP_SHARE_DEVICES") && atoi(getenv("UC_GROUP_SHARE_DEVICES")) != 0;
  for (int a = 0; a < n_devices && !share; a++)
    for (int b = a + 1; b < n_devices; b++)
      if (devices[a] == devices[b]) return fail(-EINVAL, "uc_group_create: device %d named twice", (int)devices[a]);
  int rc = load_rccl();
  if (rc) return rc;
  uc_group* g = new (std::nothrow) uc_group();
  if (!g) return fail(-ENOMEM, "uc_group_create: out of memory");
  g->world = n_devices;
  g->first_rank = 0;
  rc = group_locals(g, cfg, devices, n_devices);
  if (!rc) {
    std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
    std::vector<int> devs(devices, devices + n_devices);
    const ncclResult_t r = g_rccl.CommInitAll(comms.data(), n_devices, devs.data());
    if (r != ncclSuccess) rc = nccl_fail(r, "ncclCommInitAll");
    else
      for (int l = 0; l < n_devices; l++) g->loc[(size_t)l].comm = comms[(size_t)l];
  }
  if (rc) {
    uc_group_destroy(g);
    return rc;
  }
  *out = g;
  return 0;
}

int uc_group_create_rank(const uc_config* cfg, const void* id, int world, int rank, uc_group** out) {
  if (!cfg || !id || !out) return fail(-EINVAL, "uc_group_create_rank: NULL argument");
  *out = nullptr;
  if (world <= 0 || rank < 0 || rank >= world) return fail(-EINVAL, "uc_group_create_rank: rank %d of %d", rank, world);
  int rc = load_rccl();
  if (rc) return rc;
  uc_group* g = new (std::nothrow) uc_group();
  if (!g) return fail(-ENOMEM, "uc_group_create_rank: out of memory");
  g->world = world;
  g->first_rank = rank;
  const int32_t dev = cfg->device;
  rc = group_locals(g, cfg, &dev, 1);
  if (!rc) {
    ncclUniqueId u;
    memcpy(u.internal, id, UC_GROUP_ID_BYTES);
    const hipError_t e = hipSetDevice(dev);
    if (e != hipSuccess) rc = hip_fail(e, "hipSetDevice");
    else {
      const ncclResult_t r = g_rccl.CommInitRank(&g->loc[0].comm, world, u, rank);
      if (r != ncclSuccess) rc = nccl_fail(r, "ncclCommInitRank");
    }
  }
  if (rc) {
    uc_group_destroy(g);
    return rc;
  }
  *out = g;
  return 0;
}

// Everything uc_group_create_rank does on this rank EXCEPT entering the communicator's rendezvous: RCCL can be loaded, the
// device can be selected, a context and the group's streams can be made.  A launcher runs it on every rank and agrees on the
// results (an all-reduce of its own) BEFORE any rank calls uc_group_create_rank: a rank that would fail there never enters
// ncclCommInitRank, and the ranks that did would wait for it for ever.
int uc_group_preflight(const uc_config* cfg) {
  if (!cfg) return fail(-EINVAL, "uc_group_preflight: NULL config");
  int rc = load_rccl();
  if (rc) return rc;
  int version = 0;
  const ncclResult_t r = g_rccl.GetVersion(&version);
  if (r != ncclSuccess) return nccl_fail(r, "ncclGetVersion");
  uc_group* g = new (std::nothrow) uc_group();
  if (!g) return fail(-ENOMEM, "uc_group_preflight: out of memory");
  g->world = 1;
  g->first_rank = 0;
  const int32_t dev = cfg->device;
  rc = group_locals(g, cfg, &dev, 1);
  uc_group_destroy(g);
  return rc;
}

int uc_group_world(const uc_group* g) { return g ? g->world : fail(-EINVAL, "uc_group_world: NULL group"); }
int uc_group_local_count(const uc_group* g) { return g ? (int)g->loc.size() : fail(-EINVAL, "uc_group_local_count: NULL group"); }
int uc_group_first_rank(const uc_group* g) { return g ? g->first_rank : fail(-EINVAL, "uc_group_first_rank: NULL group"); }

uc_ctx* uc_group_ctx(uc_group* g, int local) {
  if (!g || local < 0 || (size_t)local >= g->loc.size()) {
    (void)fail(-EINVAL, "uc_group_ctx: no local device %d", local);
    return nullptr;
  }
  return g->loc[(size_t)local].ctx;
}

int uc_group_process_batch(uc_group* g, const void* const* frames, int dtype, size_t n_frames_total, size_t stride_elems,
                           uint8_t* const* gathered, void* const* hip_streams) {
  if (!g || !frames || !gathered) return fail(-EINVAL, "uc_group_process_batch: NULL argument");
  const int nl = (int)g->loc.size();
  if (n_frames_total == 0) return 0;
  const bool even = n_frames_total % (size_t)g->world == 0;
  std::vector<uint8_t*> dst((size_t)nl, nullptr);
  bool any_host = false;

  // 0. every argument of every local device is looked at BEFORE anything is enqueued: a refused call (< 0 from here) has
  //    touched no stream and started no collective -- the group is as it was, the peers are not left waiting for this rank
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32) return fail(-EINVAL, "uc_group_process_batch: bad dtype %d", dtype);
  for (int l = 0; l < nl; l++) {
    size_t first = 0, count = 0;
    uc_partition(n_frames_total, g->world, g->first_rank + l, &first, &count);
    if (!gathered[l]) return fail(-EINVAL, "uc_group_process_batch: gathered[%d] is NULL", l);
    if (count && !frames[l]) return fail(-EINVAL, "uc_group_process_batch: frames[%d] is NULL", l);
  }

  // 1. every local device decodes its shard into its slice of the gathered stream
  for (int l = 0; l < nl; l++) {
    Local& L = g->loc[(size_t)l];
    size_t first = 0, count = 0;
    uc_partition(n_frames_total, g->world, g->first_rank + l, &first, &count);
    if (!gathered[l]) return fail(-EINVAL, "uc_group_process_batch: gathered[%d] is NULL", l);
    if (count && !frames[l]) return fail(-EINVAL, "uc_group_process_batch: frames[%d] is NULL", l);
    hipError_t e = hipSetDevice(L.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipStream_t cs = (hip_streams && hip_streams[l]) ? (hipStream_t)hip_streams[l] : L.compute;
    uint8_t* d = gathered[l];
    if (!is_device_ptr(d)) {
      any_host = true;
      if (const int rc = stage_buffer(L, 0, n_frames_total, &d)) return rc;
    }
    dst[(size_t)l] = d;
    if (const int rc = hazard_wait(L, cs, d, n_frames_total, 1)) return rc;
    if (count) {
      const int rc = uc_process_batch(L.ctx, frames[l], dtype, count, stride_elems, nullptr, d + first, nullptr, cs);
      if (rc) return rc;
    }
    e = hipEventRecord(L.kernel_done, cs);
    if (e == hipSuccess) e = hipStreamWaitEvent(L.gather, L.kernel_done, 0);
    if (e != hipSuccess) return hip_fail(e, "hipEventRecord / hipStreamWaitEvent(kernel -> gather)");
  }

  // 2. the all-gather of the symbol stream, in place, all local devices inside one RCCL group
  ncclResult_t r = g_rccl.GroupStart();
  if (r != ncclSuccess) return nccl_fail(r, "ncclGroupStart");
  for (int l = 0; l < nl && r == ncclSuccess; l++) {
    Local& L = g->loc[(size_t)l];
    r = gather_in_place(g->world, g->first_rank + l, dst[(size_t)l], 1, n_frames_total, L.comm, L.gather);
  }
  const ncclResult_t r2 = g_rccl.GroupEnd();
  if (r != ncclSuccess) return nccl_fail(r, even ? "ncclAllGather" : "ncclBroadcast");
  if (r2 != ncclSuccess) return nccl_fail(r2, "ncclGroupEnd");

  // 3. remember the gather for the hazard guard; host buffers: copy the stream out and wait
  for (int l = 0; l < nl; l++) {
    Local& L = g->loc[(size_t)l];
    hipError_t e = hipSetDevice(L.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (const int rc = hazard_record(L, dst[(size_t)l], n_frames_total)) return rc;
    if (dst[(size_t)l] != gathered[l]) {
      e = hipMemcpyAsync(gathered[l], dst[(size_t)l], n_frames_total, hipMemcpyDeviceToHost, L.gather);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(gathered)");
    }
  }
  if (any_host) return uc_group_synchronize(g);
  return 0;
}

// uc_group_receive_streams / _next: "replicas across streams"  -- the streams are block-partitioned over
// the ranks, every local device runs the multi-stream receiver (ISR FIFO, dsp() at every 256-sample offset, main()'s switch
// replayed on the device: receiver/Src/main.c:417-554, 243-273, 659-668) over its share and writes texts and counts into
// its slice of the gathered arrays; both arrays are then all-gathered in place behind an event, as the symbol stream is.
static int group_receive(uc_group* g, uc_rx_state* const* states, const void* const* samples, int dtype, size_t n_streams_total,
                         size_t n_samples, size_t stream_stride_elems, const uint8_t* const* busy, char* const* text,
                         size_t text_cap, uint32_t* const* n_text, void* const* hip_streams, const char* who) {
  if (!g || !samples || !text) return fail(-EINVAL, "%s: NULL argument", who);
  if (text_cap == 0) return fail(-EINVAL, "%s: text_cap is 0", who);
  if (n_streams_total == 0) return 0;
  const int nl = (int)g->loc.size();
  const size_t text_bytes = n_streams_total * text_cap, cnt_bytes = n_streams_total * sizeof(uint32_t);
  std::vector<uint8_t*> dtext((size_t)nl, nullptr), dcnt((size_t)nl, nullptr);
  bool any_host = false;

  // every argument of every local device first: a refused call has enqueued nothing (see uc_group_process_batch)
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32 && dtype != UC_DTYPE_PDM) return fail(-EINVAL, "%s: bad dtype %d", who, dtype);
  for (int l = 0; l < nl; l++) {
    size_t first = 0, count = 0;
    uc_partition(n_streams_total, g->world, g->first_rank + l, &first, &count);
    if (!text[l]) return fail(-EINVAL, "%s: text[%d] is NULL", who, l);
    if (n_text && !n_text[l]) return fail(-EINVAL, "%s: n_text[%d] is NULL", who, l);
    // everything uc_receive_streams[_next] itself would refuse this device's share for (whole blocks, the state's dtype lock
    // and context, overlapping streams, UC_DTYPE_PDM alignment ... -- some of it differs by rank): the same function it runs
    if (count)
      if (const int rc = uc::receive_streams_check(g->loc[(size_t)l].ctx, states ? states[l] : nullptr, states != nullptr, samples[l],
                                                   dtype, count, n_samples, stream_stride_elems, text[l], text_cap, 0))
        return rc;
    if (states) {
      // (a rank that owns no stream -- fewer streams than GPUs -- has no state to bring: NULL)
      if (count && !states[l]) return fail(-EINVAL, "%s: states[%d] is NULL", who, l);
      if (uc_rx_state_streams(states[l]) != count)
        return fail(-EINVAL, "%s: states[%d] holds %zu streams, rank %d owns %zu of %zu", who, l, uc_rx_state_streams(states[l]),
                    g->first_rank + l, count, n_streams_total);
    }
  }

  for (int l = 0; l < nl; l++) {
    Local& L = g->loc[(size_t)l];
    size_t first = 0, count = 0;
    uc_partition(n_streams_total, g->world, g->first_rank + l, &first, &count);
    if (!text[l]) return fail(-EINVAL, "%s: text[%d] is NULL", who, l);
    if (states) {
      // (a rank that owns no stream -- fewer streams than GPUs -- has no state to bring: NULL)
      if (count && !states[l]) return fail(-EINVAL, "%s: states[%d] is NULL", who, l);
      if (uc_rx_state_streams(states[l]) != count)
        return fail(-EINVAL, "%s: states[%d] holds %zu streams, rank %d owns %zu of %zu", who, l, uc_rx_state_streams(states[l]),
                    g->first_rank + l, count, n_streams_total);
    }
    hipError_t e = hipSetDevice(L.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipStream_t cs = (hip_streams && hip_streams[l]) ? (hipStream_t)hip_streams[l] : L.compute;
    uint8_t* dt = (uint8_t*)text[l];
    if (!is_device_ptr(dt)) {
      any_host = true;
      if (const int rc = stage_buffer(L, 0, text_bytes, &dt)) return rc;
    }
    uint8_t* dc = nullptr;
    if (n_text) {
      if (!n_text[l]) return fail(-EINVAL, "%s: n_text[%d] is NULL", who, l);
      dc = (uint8_t*)n_text[l];
      if (!is_device_ptr(dc)) {
        any_host = true;
        if (const int rc = stage_buffer(L, 1, cnt_bytes, &dc)) return rc;
      }
    }
    dtext[(size_t)l] = dt;
    dcnt[(size_t)l] = dc;
    const unsigned n_new = dc ? 2u : 1u;
    if (const int rc = hazard_wait(L, cs, dt, text_bytes, n_new)) return rc;
    if (dc)
      if (const int rc = hazard_wait(L, cs, dc, cnt_bytes, 0)) return rc;
    if (count) {
      char* t = (char*)dt + first * text_cap;
      uint32_t* c = dc ? (uint32_t*)dc + first : nullptr;
      const uint8_t* b = (busy && busy[l]) ? busy[l] : nullptr;
      const int rc = states ? uc_receive_streams_next(L.ctx, states[l], samples[l], dtype, n_samples, stream_stride_elems, b, t,
                                                      text_cap, c, nullptr, 0, nullptr, cs)
                            : uc_receive_streams(L.ctx, samples[l], dtype, count, n_samples, stream_stride_elems, b, t, text_cap,
                                                 c, nullptr, 0, nullptr, cs);
      if (rc) return rc;
    }
    e = hipEventRecord(L.kernel_done, cs);
    if (e == hipSuccess) e = hipStreamWaitEvent(L.gather, L.kernel_done, 0);
    if (e != hipSuccess) return hip_fail(e, "hipEventRecord / hipStreamWaitEvent(receiver -> gather)");
  }

  ncclResult_t r = g_rccl.GroupStart();
  if (r != ncclSuccess) return nccl_fail(r, "ncclGroupStart");
  for (int l = 0; l < nl && r == ncclSuccess; l++) {
    Local& L = g->loc[(size_t)l];
    r = gather_in_place(g->world, g->first_rank + l, dtext[(size_t)l], text_cap, n_streams_total, L.comm, L.gather);
    if (r == ncclSuccess && dcnt[(size_t)l])
      r = gather_in_place(g->world, g->first_rank + l, dcnt[(size_t)l], sizeof(uint32_t), n_streams_total, L.comm, L.gather);
  }
  const ncclResult_t r2 = g_rccl.GroupEnd();
  if (r != ncclSuccess) return nccl_fail(r, "ncclAllGather / ncclBroadcast (texts)");
  if (r2 != ncclSuccess) return nccl_fail(r2, "ncclGroupEnd");

  for (int l = 0; l < nl; l++) {
    Local& L = g->loc[(size_t)l];
    hipError_t e = hipSetDevice(L.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (const int rc = hazard_record(L, dtext[(size_t)l], text_bytes)) return rc;
    if (dcnt[(size_t)l])
      if (const int rc = hazard_record(L, dcnt[(size_t)l], cnt_bytes)) return rc;
    if (dtext[(size_t)l] != (uint8_t*)text[l]) e = hipMemcpyAsync(text[l], dtext[(size_t)l], text_bytes, hipMemcpyDeviceToHost, L.gather);
    if (e == hipSuccess && dcnt[(size_t)l] && dcnt[(size_t)l] != (uint8_t*)n_text[l])
      e = hipMemcpyAsync(n_text[l], dcnt[(size_t)l], cnt_bytes, hipMemcpyDeviceToHost, L.gather);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(gathered texts)");
  }
  if (any_host) return uc_group_synchronize(g);
  return 0;
}

int uc_group_receive_streams(uc_group* g, const void* const* samples, int dtype, size_t n_streams_total, size_t n_samples,
                             size_t stream_stride_elems, const uint8_t* const* busy, char* const* text, size_t text_cap,
                             uint32_t* const* n_text, void* const* hip_streams) {
  return group_receive(g, nullptr, samples, dtype, n_streams_total, n_samples, stream_stride_elems, busy, text, text_cap, n_text,
                       hip_streams, "uc_group_receive_streams");
}

int uc_group_receive_streams_next(uc_group* g, uc_rx_state* const* states, const void* const* samples, int dtype,
                                  size_t n_streams_total, size_t n_samples, size_t stream_stride_elems,
                                  const uint8_t* const* busy, char* const* text, size_t text_cap, uint32_t* const* n_text,
                                  void* const* hip_streams) {
  if (!states) return fail(-EINVAL, "uc_group_receive_streams_next: states is NULL");
  return group_receive(g, states, samples, dtype, n_streams_total, n_samples, stream_stride_elems, busy, text, text_cap, n_text,
                       hip_streams, "uc_group_receive_streams_next");
}

// uc_group_process_stream: UC_STREAM over the GPUs of a node.  The overlap-save BLOCKS of a stream are independent, so the
// stream shards by whole blocks (uc_stream_span: block boundaries as in the one-GPU run, every shard re-reads the `halo`
// samples in front of it -- read-only duplication, no exchange); the compressed envelope stays where it was computed
// (4 / D bytes per input sample: nothing anybody wants on every GPU), the per-block peak records (8 bytes per block) are
// all-gathered in place like the symbol stream.
int uc_group_process_stream(uc_group* g, const void* const* samples, int dtype, size_t n_samples_total, float* const* compressed,
                            uc_peak* const* peaks, void* const* hip_streams) {
  if (!g || !samples || !peaks) return fail(-EINVAL, "uc_group_process_stream: NULL argument");
  const int nl = (int)g->loc.size();
  size_t n_out_total = 0, n_blocks = 0, hop = 0;
  if (const int rc = uc_stream_geometry(g->loc[0].ctx, n_samples_total, nullptr, &n_out_total, &n_blocks, &hop)) return rc;
  if (n_blocks == 0) return 0;
  const size_t peak_bytes = n_blocks * sizeof(uc_peak);
  std::vector<uint8_t*> dst((size_t)nl, nullptr);
  bool any_host = false;
  // every argument of every local device first: a refused call has enqueued nothing (see uc_group_process_batch)
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32) return fail(-EINVAL, "uc_group_process_stream: bad dtype %d", dtype);
  for (int l = 0; l < nl; l++) {
    size_t first_sample = 0, n_shard = 0, first_out = 0, n_out = 0;
    if (const int rc = uc_stream_span(g->loc[(size_t)l].ctx, n_samples_total, g->world, g->first_rank + l, &first_sample, &n_shard,
                                      &first_out, &n_out))
      return rc;
    if (!peaks[l]) return fail(-EINVAL, "uc_group_process_stream: peaks[%d] is NULL", l);
    if (n_out && !samples[l]) return fail(-EINVAL, "uc_group_process_stream: samples[%d] is NULL", l);
  }
  for (int l = 0; l < nl; l++) {
    Local& L = g->loc[(size_t)l];
    size_t first_sample = 0, n_shard = 0, first_out = 0, n_out = 0;
    if (const int rc = uc_stream_span(L.ctx, n_samples_total, g->world, g->first_rank + l, &first_sample, &n_shard, &first_out, &n_out))
      return rc;
    if (!peaks[l]) return fail(-EINVAL, "uc_group_process_stream: peaks[%d] is NULL", l);
    if (n_out && !samples[l]) return fail(-EINVAL, "uc_group_process_stream: samples[%d] is NULL", l);
    hipError_t e = hipSetDevice(L.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    hipStream_t cs = (hip_streams && hip_streams[l]) ? (hipStream_t)hip_streams[l] : L.compute;
    uint8_t* d = (uint8_t*)peaks[l];
    if (!is_device_ptr(d)) {
      any_host = true;
      if (const int rc = stage_buffer(L, 0, peak_bytes, &d)) return rc;
    }
    dst[(size_t)l] = d;
    if (const int rc = hazard_wait(L, cs, d, peak_bytes, 1)) return rc;
    if (n_out) {
      // (uc_stream_span cuts at multiples of the hop: the shard's first block is block first_out / hop of the stream)
      uc_peak* mine = (uc_peak*)d + first_out / hop;
      float* comp = (compressed && compressed[l]) ? compressed[l] : nullptr;
      if (const int rc = uc_process_stream(L.ctx, samples[l], dtype, n_shard, comp, mine, cs)) return rc;
    }
    e = hipEventRecord(L.kernel_done, cs);
    if (e == hipSuccess) e = hipStreamWaitEvent(L.gather, L.kernel_done, 0);
    if (e != hipSuccess) return hip_fail(e, "hipEventRecord / hipStreamWaitEvent(stream kernel -> gather)");
  }
  ncclResult_t r = g_rccl.GroupStart();
  if (r != ncclSuccess) return nccl_fail(r, "ncclGroupStart");
  for (int l = 0; l < nl && r == ncclSuccess; l++) {
    Local& L = g->loc[(size_t)l];
    r = gather_in_place(g->world, g->first_rank + l, dst[(size_t)l], sizeof(uc_peak), n_blocks, L.comm, L.gather);
  }
  const ncclResult_t r2 = g_rccl.GroupEnd();
  if (r != ncclSuccess) return nccl_fail(r, "ncclAllGather / ncclBroadcast (peaks)");
  if (r2 != ncclSuccess) return nccl_fail(r2, "ncclGroupEnd");
  for (int l = 0; l < nl; l++) {
    Local& L = g->loc[(size_t)l];
    hipError_t e = hipSetDevice(L.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    if (const int rc = hazard_record(L, dst[(size_t)l], peak_bytes)) return rc;
    if (dst[(size_t)l] != (uint8_t*)peaks[l]) {
      e = hipMemcpyAsync(peaks[l], dst[(size_t)l], peak_bytes, hipMemcpyDeviceToHost, L.gather);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(gathered peaks)");
    }
  }
  if (any_host) return uc_group_synchronize(g);
  return 0;
}

int uc_group_wait_gather(uc_group* g, int local, const uint8_t* gathered, void* hip_stream) {
  if (!g || local < 0 || (size_t)local >= g->loc.size() || !gathered)
    return fail(-EINVAL, "uc_group_wait_gather: bad argument");
  Local& L = g->loc[(size_t)local];
  hipError_t e = hipSetDevice(L.device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  // the most recent gather into that buffer
  for (unsigned back = 1; back <= (unsigned)kHazardRing && back <= L.hz_next; back++) {
    const unsigned k = (L.hz_next - back) % kHazardRing;
    if (L.hz_buf[k] == gathered) {
      e = hipStreamWaitEvent((hipStream_t)hip_stream, L.hz_ev[k], 0);
      if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent(gather -> caller)");
      return 0;
    }
  }
  return fail(-ENOENT, "uc_group_wait_gather: no recent gather into that buffer");
}

int uc_group_synchronize(uc_group* g) {
  if (!g) return fail(-EINVAL, "uc_group_synchronize: NULL group");
  for (Local& L : g->loc) {
    hipError_t e = hipSetDevice(L.device);
    if (e == hipSuccess) e = hipStreamSynchronize(L.gather);
    if (e == hipSuccess) e = hipStreamSynchronize(L.compute);
    if (e != hipSuccess) return hip_fail(e, "uc_group_synchronize");
  }
  return 0;
}

}  // extern "C"
