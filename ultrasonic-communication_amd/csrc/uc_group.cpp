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

namespace uc {
void set_error(const char* msg);  // uc_api.cpp: the thread's uc_last_error() text
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
  void* d_gathered = nullptr;  // host-pointer calls: the stream is gathered here, then copied out
  size_t d_gathered_cap = 0;
};

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
    if (L.d_gathered) (void)hipFree(L.d_gathered);
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
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&L.gather, hipStreamNonBlocking);
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
  if (cfg->variant == UC_STREAM) return fail(-ENOTSUP, "uc_group_create: UC_STREAM has no frames (shard it with uc_stream_span)");
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
  if (cfg->variant == UC_STREAM) return fail(-ENOTSUP, "uc_group_create_rank: UC_STREAM has no frames (shard it with uc_stream_span)");
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

static bool overlaps(const uint8_t* a, size_t na, const uint8_t* b, size_t nb) { return a < b + nb && b < a + na; }

int uc_group_process_batch(uc_group* g, const void* const* frames, int dtype, size_t n_frames_total, size_t stride_elems,
                           uint8_t* const* gathered, void* const* hip_streams) {
  if (!g || !frames || !gathered) return fail(-EINVAL, "uc_group_process_batch: NULL argument");
  const int nl = (int)g->loc.size();
  if (n_frames_total == 0) return 0;
  const bool even = n_frames_total % (size_t)g->world == 0;
  std::vector<uint8_t*> dst((size_t)nl, nullptr);
  bool any_host = false;

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
      if (L.d_gathered_cap < n_frames_total) {
        if (L.d_gathered) (void)hipFree(L.d_gathered);
        L.d_gathered = nullptr;
        L.d_gathered_cap = 0;
        e = hipMalloc(&L.d_gathered, n_frames_total);
        if (e != hipSuccess) return hip_fail(e, "hipMalloc(gathered)");
        L.d_gathered_cap = n_frames_total;
      }
      d = (uint8_t*)L.d_gathered;
    }
    dst[(size_t)l] = d;
    // write-after-gather: an earlier gather that still reads or writes this buffer must be done before the kernel
    // overwrites the rank's slice of it (device-side wait, nothing blocks here)
    // ... and the ring slot this step will recycle: a gather that drops out of the ring must be complete before anything
    // newer runs, or a caller rotating more than kHazardRing buffers could overwrite one behind the guard's back
    const unsigned recycle = L.hz_next % kHazardRing;
    for (int k = 0; k < kHazardRing; k++)
      if (L.hz_buf[k] && ((unsigned)k == recycle || overlaps(L.hz_buf[k], L.hz_bytes[k], d, n_frames_total))) {
        e = hipStreamWaitEvent(cs, L.hz_ev[k], 0);
        if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent(gather -> kernel)");
      }
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
    uint8_t* d = dst[(size_t)l];
    if (even) {
      const size_t per = n_frames_total / (size_t)g->world;
      r = g_rccl.AllGather(d + (size_t)(g->first_rank + l) * per, d, per, ncclUint8, L.comm, L.gather);
    } else {
      // ragged shares: one broadcast per rank, each slice from its owner
      for (int root = 0; root < g->world && r == ncclSuccess; root++) {
        size_t first = 0, count = 0;
        uc_partition(n_frames_total, g->world, root, &first, &count);
        if (count) r = g_rccl.Broadcast(d + first, d + first, count, ncclUint8, root, L.comm, L.gather);
      }
    }
  }
  const ncclResult_t r2 = g_rccl.GroupEnd();
  if (r != ncclSuccess) return nccl_fail(r, even ? "ncclAllGather" : "ncclBroadcast");
  if (r2 != ncclSuccess) return nccl_fail(r2, "ncclGroupEnd");

  // 3. remember the gather for the hazard guard; host buffers: copy the stream out and wait
  for (int l = 0; l < nl; l++) {
    Local& L = g->loc[(size_t)l];
    hipError_t e = hipSetDevice(L.device);
    if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
    const unsigned k = L.hz_next++ % kHazardRing;
    e = hipEventRecord(L.hz_ev[k], L.gather);
    if (e != hipSuccess) return hip_fail(e, "hipEventRecord(gather)");
    L.hz_buf[k] = dst[(size_t)l];
    L.hz_bytes[k] = n_frames_total;
    if (dst[(size_t)l] != gathered[l]) {
      e = hipMemcpyAsync(gathered[l], dst[(size_t)l], n_frames_total, hipMemcpyDeviceToHost, L.gather);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(gathered)");
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
