// uc_api.cpp -- the C-ABI of include/uchirp.h on top of the gfx950 kernels.
// No CPU compute path exists here: without a usable HIP device uc_create fails.
#include <errno.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/uchirp.h"
#include "../../include/uchirp_mainloop.hpp"
#include "uc_kernels.hpp"
#include "uc_rx.hpp"
#include "uc_tables.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int hip_fail(hipError_t e, const char* what) {
  return fail(-EIO, "%s: %s", what, hipGetErrorString(e));
}

// set while a call records into a stream capture: scratch must not be (re)allocated there (hipMalloc / hipFree are not
// capturable, and a freed buffer may be baked into the graph) -- a buffer that would have to grow fails the call instead
thread_local bool g_capturing = false;
struct CaptureNoAlloc {
  const bool prev;
  explicit CaptureNoAlloc(bool on) : prev(g_capturing) { g_capturing = on || prev; }
  ~CaptureNoAlloc() { g_capturing = prev; }
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return 0;
    if (g_capturing)
      return fail(-ENOBUFS, "scratch of %zu bytes would have to be allocated during a stream capture: make one eager call of "
                            "the same shape first (it sizes the scratch)", bytes);
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return hip_fail(e, "hipMalloc(staging)");
    cap = want;
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// (Not cached: device and host allocations share one virtual address space, so an address that was device memory can
// later be host memory; a stale "device" verdict would hand a host pointer to a kernel.  The query costs a few
// microseconds per pointer argument, which only small batches notice.)
bool is_device_ptr(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  memset(&attr, 0, sizeof(attr));
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // clear the sticky "invalid value" of a plain host pointer
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// what one uc_receive_streams[_next] call parks on the device between its kernels: staged host inputs, the accepted-block
// lists and the packed copy of a busy-masked call, the new (up, down) records, staged host outputs.  A live state owns a set
// of its own (calls on different states never share scratch); calls without a state use the context's, one stream at a time
// (rx_guard below).
struct RxScratch {
  DevBuf in, busy, acc, na, pad, rec, text, ntext, trace, ntrace;
  DevBuf pcm, hist;  // UC_DTYPE_PDM: the DFSDM words of the call's blocks; the filter history of streams without a state
  void release() {
    for (DevBuf* b : {&in, &busy, &acc, &na, &pad, &rec, &text, &ntext, &trace, &ntrace, &pcm, &hist}) b->release();
  }
};

}  // namespace

namespace uc {
void set_error(const char* msg) { g_err = msg ? msg : ""; }  // (uc_group.cpp reports through the same uc_last_error())
}

// Hand-out counters of the dynamically dealt launches (two 32-bit words each -- next ticket, workgroups gone -- every
// pair in its own 128-byte line; zeroed once at uc_create, left at zero by every launch's last workgroup):
//   slots [0, kWorkSlots)                          a ring for EAGER launches; a slot is reused only after the launch that
//                                                  last used it has finished (one hipEvent per slot, queried on reuse)
//   slots [kWorkSlots, kWorkSlots + kGraphSlots)   handed out ONCE each to launches recorded while their stream is being
//                                                  captured into a hipGraph: the graph owns that slot for the life of the
//                                                  context
constexpr unsigned kWorkSlots = 64, kGraphSlots = 960, kWorkStride = 128;

struct uc_ctx {
  uc_config cfg;
  uc::Tables tab;
  uc::StreamTables stab;  // UC_STREAM only
  int device = 0;
  int num_cu = 256;
  // device-resident tables
  float2* d_tab0 = nullptr;
  float2* d_tab1 = nullptr;
  float2* d_tab2 = nullptr;  // IQ base band: conj(down chirp) * hann
  float2* d_tw = nullptr;
  float* d_aux = nullptr;  // variant-specific (COMPRESS: H_down packed; IQ: carrier/fir/...)
  int32_t* d_cic4 = nullptr;  // sinc^5 byte tables, built on first use of uc_dfsdm_sinc5
  int32_t* d_cic1 = nullptr;
  int cic_blocks_per_cu = 0;
  DevBuf s_cic_in, s_cic_out, s_cic_hist;
  // staging for host-pointer calls
  DevBuf s_frames, s_mm, s_sym, s_stats;
  int band_blocks_per_cu[6][3][2] = {};  // [default / wide / default + spectrum stores / rows / rows + wide / overlapping frames][mode][dtype]: the instantiations differ in registers
  int full_blocks_per_cu[2] = {0, 0};    // [dtype]: the int32 / f32 instantiations differ in registers
  int iq_blocks_per_cu[2] = {0, 0};
  int stream_blocks_per_cu[2] = {0, 0};
  DevBuf s_comp, s_peaks, s_spec;
  DevBuf s_rx_pad, s_rx_mag;        // uc_receive_stream: the zero-prefixed stream, (up, down) mag_max per frame
  RxScratch rx;                     // uc_receive_streams (no live state): scratch of the call in flight
  void* d_zero_block = nullptr;     // n zero words: the block "in front of" a stream that starts (fifo_queue at power-on,
                                    // main.c:94) and the zero records such a stream carries in
  hipEvent_t rx_ev = nullptr;       // recorded behind the last uc_receive_streams call that used `rx`: a call on ANOTHER
  hipStream_t rx_stream = nullptr;  // stream waits for it (on the device) before it overwrites the scratch
  bool rx_used = false;
  std::vector<float2> h_rx_mag;
  int band_waves = 3;     // tuning knobs (env UC_BAND_WAVES / UC_GRID / UC_BAND_GROUP / UC_STATIC_DEAL): not part of the ABI
  bool band_waves_set = false;  // UC_BAND_WAVES given: use it for every mode (default: 3, SYNC_CPLX 2 -- see process_batch_impl)
  int grid_override = 0;
  int band_group = 32;    // frames per group handed to a workgroup at a time
  bool static_deal = false;
  uint32_t rx_need_force = 0;  // (env UC_RX_NEED_FORCE=0x1..: pricing runs only) every stream's need word is this one: WRONG results
  bool rx_poison = false;   // (env UC_RX_POISON=1, tests) the statistics the live receivers pass over are huge instead of zero
  int compress_chunk = 8;   // (env UC_COMPRESS_CHUNK) frame pairs per hand-out chunk of the compress kernel: a power of two >= 2
  int stream_chunk = 2;     // (env UC_STREAM_CHUNK) blocks per hand-out chunk of the stream kernel: a power of two
  int iq_group = 16;        // (env UC_IQ_GROUP) frames per hand-out group of the IQ kernels: a power of two <= 64
                            // (16: +0.6 ... 1.2 % over 32 on all three IQ kernels, profiles/r04_knob_sweep.txt)
  unsigned iq_stagger = 0;  // (env UC_IQ_STAGGER, MFMA FIR only) start delay of every second wave on a SIMD, x 4096 clocks
  // UC_IQ at n = 1024, env UC_IQ_FIR=mfma: the FIR as v_mfma_f32_16x16x4_f32 Toeplitz tiles instead of packed VALU.
  // Off by default: an f32 MFMA and the partner wave's packed-f32 VALU do not overlap on a SIMD (tools/mfma_valu_probe.hip:
  // together they take the SUM of their times), and the Toeplitz padding makes the matrix form 1.6x the FMA count.
  bool iq_fir_mfma = false;
  // work counters for the dynamic group hand-out: one word per launch, a ring so that launches of one context that
  // overlap on different streams never share one (each word sits in its own 128-byte line)
  unsigned int* d_work = nullptr;
  void* h_slot = nullptr;  // uc_process_frame: pinned, device-mapped host memory for one frame and its results
  unsigned work_next = 0;
  unsigned graph_next = 0;              // graph-owned slots handed out so far (never recycled)
  hipEvent_t work_ev[kWorkSlots] = {};  // recorded behind the launch that used ring slot i (several streams only)
  bool work_busy[kWorkSlots] = {};      // slot i has been used and its event not yet seen complete
  // As long as every eager launch of the context goes to ONE stream, stream order alone keeps a slot from being
  // shared (its previous user finished 64 launches earlier on the same stream) and no event is recorded at all.
  // The first launch on a second stream records `switch_ev` on the first one -- it covers every slot used so far --
  // and from then on every launch records its slot's event.
  hipStream_t ring_stream = nullptr;
  bool ring_stream_set = false, multi_stream = false;
  hipEvent_t switch_ev = nullptr;
  bool wait_switch[kWorkSlots] = {};    // slot i was last used before the switch: free once switch_ev has completed
  bool slot_used[kWorkSlots] = {};
  bool switch_lost = false;             // the switch event could not be recorded: slots used before it never come back
  bool graph_slots_warned = false;
  // uc_clock_probe(): launches run the clock-stamped twin of their kernel (uc_kernels.hpp: uc::clk) and leave four words
  // per wave here; clock_waves = the waves of the LAST launch
  bool clock_probe = false;
  DevBuf s_clock;
  size_t clock_waves = 0;
  int clk_cic_blocks = 0;
};

extern "C" {

int uc_abi_version(void) { return UC_ABI_VERSION; }

const char* uc_last_error(void) { return g_err.c_str(); }

int uc_default_config(int32_t variant, uc_config* cfg) {
  if (!cfg) return fail(-EINVAL, "uc_default_config: cfg is NULL");
  memset(cfg, 0, sizeof(*cfg));
  cfg->n = 2048;              // receiver/Inc/main.h:97
  cfg->phase_deg = -90.0f;    // receiver/Src/chirp.c:43-44
  cfg->snr_threshold = 2.0f;  // receiver/Inc/main.h:98
  cfg->mag_mean = 1.0f;
  cfg->carrier = 18000.0f;    // experiments/iq_modulation/Inc/iq_modem.h:10
  cfg->variant = variant;
  switch (variant) {
    case UC_RX_REAL:
    case UC_SYNC_CPLX:
      cfg->fs = 78125.0f;  // 80 MHz / 32 / 32 / 1: receiver/Src/main.c:367-369, dfsdm.c:60-61,69
      cfg->f0 = 16000.0f;  // receiver/Inc/chirp.h:18-19
      cfg->f1 = 19000.0f;
      cfg->time_frame = 0.0205f;  // receiver/Inc/chirp.h:16
      return 0;
    case UC_COMPRESS:
    case UC_DECHIRP_DOWN:
      cfg->fs = 100000.0f;  // Divider 25: experiments/chirp_compression_*/Src/dfsdm.c:73
      cfg->f0 = 17000.0f;   // experiments/chirp_compression_*/Inc/chirp.h (F1, F2)
      cfg->f1 = 18000.0f;
      cfg->time_frame = 0.0f;  // n / fs
      return 0;
    case UC_IQ:
      cfg->fs = 100000.0f;
      cfg->f0 = 16000.0f;
      cfg->f1 = 19000.0f;
      cfg->time_frame = 0.0205f;
      return 0;
    case UC_STREAM:  // the shipping receiver's band and rate, carrier at the band centre
      cfg->fs = 78125.0f;
      cfg->f0 = 16000.0f;
      cfg->f1 = 19000.0f;
      cfg->time_frame = 0.0f;  // one symbol = n samples
      cfg->carrier = 17500.0f;
      cfg->decim = 8;
      return 0;
    default:
      return fail(-EINVAL, "uc_default_config: unknown variant %d", (int)variant);
  }
}

// (a table that already lives on the device is overwritten in place: uc_set_table)
static int upload(void** dst, const void* src, size_t bytes) {
  hipError_t e = hipSuccess;
  if (!*dst) e = hipMalloc(dst, bytes);
  if (e != hipSuccess) return hip_fail(e, "hipMalloc(table)");
  e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(table)");
  return 0;
}

// The fused device tables of a context (reference * Hann etc.), from the host tables in c->tab / c->stab.
// Called by uc_create and again by uc_set_table.
static int upload_device_tables(uc_ctx* c) {
  const uc_config* cfg = &c->cfg;
  const uint32_t n = cfg->n;
  const bool iq_bb = cfg->variant == UC_IQ && (cfg->flags & UC_FLAG_IQ_BASEBAND) != 0;
  std::vector<float> t0(2 * (size_t)n, 0.0f), t1(2 * (size_t)n, 0.0f), t2;
  const uc::Tables& T = c->tab;
  switch (cfg->variant) {
    case UC_RX_REAL:
      for (uint32_t i = 0; i < n; i++) {
        t0[2 * i] = T.up[i] * T.hann[i];
        t0[2 * i + 1] = T.down[i] * T.hann[i];
      }
      break;
    case UC_DECHIRP_DOWN:
      for (uint32_t i = 0; i < n; i++) t0[2 * i] = T.down[i] * T.hann[i];
      break;
    case UC_SYNC_CPLX:
      for (uint32_t i = 0; i < n; i++) {
        t0[2 * i] = T.up[2 * i] * T.hann[i];
        t0[2 * i + 1] = T.up[2 * i + 1] * T.hann[i];
        t1[2 * i] = T.down[2 * i] * T.hann[i];
        t1[2 * i + 1] = T.down[2 * i + 1] * T.hann[i];
      }
      break;
    case UC_COMPRESS: {
      // t0 <- full Hermitian spectrum of the windowed down chirp, scaled by 1/n (the inverse
      // RFFT's scaling, chirp_compression_time_domain/Src/chirp.c:82); t1[0..n) <- symmetric Hann
      const std::vector<float>& pk = T.h_down;
      const float inv = 1.0f / (float)n;
      t0[0] = pk[0] * inv;
      t0[1] = 0.0f;
      t0[2 * (n / 2)] = pk[1] * inv;
      t0[2 * (n / 2) + 1] = 0.0f;
      for (uint32_t k = 1; k < n / 2; k++) {
        t0[2 * k] = pk[2 * k] * inv;
        t0[2 * k + 1] = pk[2 * k + 1] * inv;
        t0[2 * (n - k)] = pk[2 * k] * inv;
        t0[2 * (n - k) + 1] = -pk[2 * k + 1] * inv;
      }
      for (uint32_t i = 0; i < n; i++) t1[i] = T.hann[i];
      break;
    }
    case UC_IQ:
      // t0 <- carrier (cos, sin); t1 <- down chirp (cos, sin) * hann (Hann duplicated per re/im,
      // experiments/iq_modulation/Src/main.c:126,237)
      for (uint32_t i = 0; i < n; i++) {
        t0[2 * i] = T.carrier_c[i];
        t0[2 * i + 1] = T.carrier_s[i];
        t1[2 * i] = T.down[2 * i] * T.hann[i];
        t1[2 * i + 1] = T.down[2 * i + 1] * T.hann[i];
      }
      if (iq_bb) {
        // R * chirp.conjugate() (IQ_modulation.ipynb cells 29, 30): t1 <- conj(up) * hann, t2 <- conj(down) * hann
        t2.resize(2 * (size_t)n);
        for (uint32_t i = 0; i < n; i++) {
          t1[2 * i] = T.up[2 * i] * T.hann[i];
          t1[2 * i + 1] = -(T.up[2 * i + 1] * T.hann[i]);
          t2[2 * i] = T.down[2 * i] * T.hann[i];
          t2[2 * i + 1] = -(T.down[2 * i + 1] * T.hann[i]);
        }
      }
      break;
    case UC_STREAM:
      // t0 <- H/n (spectrum of the zero-padded template), t1 <- per-sample carrier rotation
      t0 = c->stab.hn;
      t1 = c->stab.rot;
      break;
    default:
      break;
  }
  int rc = upload((void**)&c->d_tab0, t0.data(), t0.size() * sizeof(float));
  if (!rc) rc = upload((void**)&c->d_tab1, t1.data(), t1.size() * sizeof(float));
  if (!rc && !t2.empty()) rc = upload((void**)&c->d_tab2, t2.data(), t2.size() * sizeof(float));
  return rc;
}

int uc_create(const uc_config* cfg, uc_ctx** out) {
  if (!cfg || !out) return fail(-EINVAL, "uc_create: NULL argument");
  *out = nullptr;
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return fail(-ENODEV, "uc_create: no HIP device (%s); this library has no CPU path",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  }
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(-ENODEV, "uc_create: device %d out of range [0,%d)", (int)cfg->device, ndev);
  if (cfg->n != (uint32_t)uc::kN && !(cfg->variant == UC_IQ && cfg->n == 1024))
    return fail(-ENOTSUP, "uc_create: n=%u unsupported (kernels are specialised for n=%d; UC_IQ also takes 1024)",
                cfg->n, uc::kN);

  uc_ctx* c = new (std::nothrow) uc_ctx();
  if (!c) return fail(-ENOMEM, "uc_create: out of memory");
  c->cfg = *cfg;
  c->device = cfg->device;
  // Experiment switches (grid size, group sizes, static deal, kernel variants): NOT part of the ABI.  They are read only
  // when UC_TUNING=1 is set as well, so that a stray variable in a production environment changes nothing.
  const char* tuning = getenv("UC_TUNING");
  if (tuning && atoi(tuning) != 0) {
    if (const char* w = getenv("UC_BAND_WAVES")) {
      const int v = atoi(w);
      if (v >= 2 && v <= 4) { c->band_waves = v; c->band_waves_set = true; }
    }
    if (const char* g = getenv("UC_GRID")) c->grid_override = atoi(g);
    if (const char* g = getenv("UC_RX_POISON")) c->rx_poison = atoi(g) != 0;
    if (const char* g = getenv("UC_RX_NEED_FORCE")) c->rx_need_force = (uint32_t)strtoul(g, nullptr, 0) | 0x80000000u;
    if (const char* g = getenv("UC_BAND_GROUP")) {
      const int v = atoi(g);
      if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) c->band_group = v;
    }
    if (const char* g = getenv("UC_STATIC_DEAL")) c->static_deal = atoi(g) != 0;
    if (const char* g = getenv("UC_SLOT_EVENTS")) c->multi_stream = atoi(g) != 0;  // record an event behind every launch
    if (const char* g = getenv("UC_IQ_FIR")) c->iq_fir_mfma = strcmp(g, "mfma") == 0;
    if (const char* g = getenv("UC_IQ_STAGGER")) c->iq_stagger = (unsigned)atoi(g);
    if (const char* g = getenv("UC_COMPRESS_CHUNK")) {
      const int v = atoi(g);
      if (v >= 2 && v <= 64 && (v & (v - 1)) == 0) c->compress_chunk = v;
    }
    if (const char* g = getenv("UC_STREAM_CHUNK")) {
      const int v = atoi(g);
      if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) c->stream_chunk = v;
    }
    if (const char* g = getenv("UC_IQ_GROUP")) {
      const int v = atoi(g);
      if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) c->iq_group = v;
    }
  }
  int rc = uc::build_tables(*cfg, c->tab);
  if (rc) {
    delete c;
    return fail(rc, "uc_create: invalid configuration (rc=%d)", rc);
  }
  if (cfg->variant == UC_STREAM) {
    rc = uc::build_stream_tables(*cfg, c->stab);
    if (rc) {
      delete c;
      return fail(rc, "uc_create: UC_STREAM takes decim 4, 8 or 16 (got %u)", cfg->decim);
    }
    c->cfg.decim = c->stab.decim;
  }
  if (cfg->variant != UC_IQ && cfg->variant != UC_COMPRESS && cfg->variant != UC_STREAM &&
      c->tab.bandwidth2 > (uint32_t)uc::kBandWideMax) {
    delete c;
    return fail(-ENOTSUP, "uc_create: bandwidth2=%u exceeds the %d-bin window the kernel evaluates",
                c->tab.bandwidth2, uc::kBandWideMax);
  }
  if (cfg->variant == UC_IQ && c->tab.bandwidth4 > (cfg->n == 1024 ? 128u : 256u)) {
    delete c;
    return fail(-ENOTSUP, "uc_create: IQ window of %u bins exceeds the 256 the kernel evaluates", c->tab.bandwidth4);
  }
  e = hipSetDevice(c->device);
  if (e != hipSuccess) {
    delete c;
    return hip_fail(e, "hipSetDevice");
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c->device) == hipSuccess && prop.multiProcessorCount > 0)
    c->num_cu = prop.multiProcessorCount;

  const uint32_t n = cfg->n;
  std::vector<float> tw;
  uc::build_twiddles(n, tw);
  rc = upload((void**)&c->d_tw, tw.data(), tw.size() * sizeof(float));

  if (!rc) {
    e = hipMalloc((void**)&c->d_work, (size_t)(kWorkSlots + kGraphSlots) * kWorkStride);
    if (e != hipSuccess) rc = hip_fail(e, "hipMalloc(work counters)");
    if (!rc) {
      e = hipMemset(c->d_work, 0, (size_t)(kWorkSlots + kGraphSlots) * kWorkStride);
      if (e != hipSuccess) rc = hip_fail(e, "hipMemset(work counters)");
    }
    for (unsigned i = 0; !rc && i < kWorkSlots; i++) {
      e = hipEventCreateWithFlags(&c->work_ev[i], hipEventDisableTiming);
      if (e != hipSuccess) { c->work_ev[i] = nullptr; rc = hip_fail(e, "hipEventCreate(work counter)"); }
    }
    if (!rc) {
      e = hipEventCreateWithFlags(&c->switch_ev, hipEventDisableTiming);
      if (e != hipSuccess) { c->switch_ev = nullptr; rc = hip_fail(e, "hipEventCreate(work counter)"); }
    }
  }
  if (!rc) rc = upload_device_tables(c);
  if (!rc && (cfg->variant == UC_RX_REAL || cfg->variant == UC_SYNC_CPLX)) {
    e = hipMalloc(&c->d_zero_block, (size_t)n * 4);
    if (e == hipSuccess) e = hipMemset(c->d_zero_block, 0, (size_t)n * 4);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->rx_ev, hipEventDisableTiming);
    if (e != hipSuccess) rc = hip_fail(e, "uc_create: receiver scratch");
  }
  if (!rc && cfg->variant == UC_IQ) {
    // the taps as the A operand of v_mfma_f32_16x16x4_f32: lane l = (k = l >> 4, i = l & 15) of k-step s holds
    // T[i][4 s + k] = fir[i + 26 - (4 s + k)] (0 outside the taps): output i of a 16-output block sees the
    // samples i .. i + 26 of the block's 42-sample window
    std::vector<float> fa(11 * 64, 0.0f);
    for (int s = 0; s < 11; s++)
      for (int l = 0; l < 64; l++) {
        const int d = (l & 15) + 26 - (4 * s + (l >> 4));
        if (d >= 0 && d < uc::kFirTaps) fa[(size_t)s * 64 + l] = c->tab.fir[(size_t)d];
      }
    rc = upload((void**)&c->d_aux, fa.data(), fa.size() * sizeof(float));
  }
  if (rc) {
    uc_destroy(c);
    return rc;
  }
  *out = c;
  return 0;
}

void uc_destroy(uc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_work) {
    const char* tuning = getenv("UC_TUNING");
    if (tuning && atoi(tuning) != 0) {  // debug / CI: a counter left non-zero = a kernel exit that skipped handout_leave
      const int busy = uc_debug_busy_counters(c);
      if (busy > 0) fprintf(stderr, "libuchirp: uc_destroy(%p): %d hand-out counter word(s) not zero\n", (void*)c, busy);
    }
  }
  if (c->d_tab0) (void)hipFree(c->d_tab0);
  if (c->d_tab1) (void)hipFree(c->d_tab1);
  if (c->d_tab2) (void)hipFree(c->d_tab2);
  if (c->d_tw) (void)hipFree(c->d_tw);
  if (c->d_work) (void)hipFree(c->d_work);
  for (unsigned i = 0; i < kWorkSlots; i++)
    if (c->work_ev[i]) (void)hipEventDestroy(c->work_ev[i]);
  if (c->switch_ev) (void)hipEventDestroy(c->switch_ev);
  if (c->h_slot) (void)hipHostFree(c->h_slot);
  if (c->d_aux) (void)hipFree(c->d_aux);
  if (c->d_cic4) (void)hipFree(c->d_cic4);
  if (c->d_cic1) (void)hipFree(c->d_cic1);
  c->s_cic_in.release();
  c->s_cic_out.release();
  c->s_cic_hist.release();
  c->s_frames.release();
  c->s_mm.release();
  c->s_sym.release();
  c->s_stats.release();
  c->s_comp.release();
  c->s_peaks.release();
  c->s_spec.release();
  c->s_rx_pad.release();
  c->s_rx_mag.release();
  c->rx.release();
  if (c->d_zero_block) (void)hipFree(c->d_zero_block);
  if (c->rx_ev) (void)hipEventDestroy(c->rx_ev);
  c->s_clock.release();
  delete c;
}

static bool iq_baseband(const uc_ctx* c) {
  return c->cfg.variant == UC_IQ && (c->cfg.flags & UC_FLAG_IQ_BASEBAND) != 0;
}

int uc_stats_per_frame(const uc_ctx* c) {
  if (!c) return fail(-EINVAL, "uc_stats_per_frame: NULL ctx");
  return (c->cfg.variant == UC_RX_REAL || c->cfg.variant == UC_SYNC_CPLX || iq_baseband(c)) ? 2 : 1;
}

int uc_iq_halo(const uc_ctx* c) {
  if (!c) return fail(-EINVAL, "uc_iq_halo: NULL ctx");
  return c->cfg.variant == UC_IQ ? uc::kFirTaps - 1 : 0;
}

int uc_get_windows(const uc_ctx* c, uint32_t* bw, uint32_t* bw2, uint32_t* ilz) {
  if (!c) return fail(-EINVAL, "uc_get_windows: NULL ctx");
  if (bw) *bw = c->tab.bandwidth;
  if (bw2) *bw2 = c->tab.bandwidth2;
  if (ilz) *ilz = c->tab.idx_left_zero;
  return 0;
}

int uc_get_table(const uc_ctx* c, int id, float* out, size_t cap) {
  if (!c || !out) return fail(-EINVAL, "uc_get_table: NULL argument");
  const std::vector<float>* v = nullptr;
  switch (id) {
    case UC_TABLE_UP: v = &c->tab.up; break;
    case UC_TABLE_DOWN: v = &c->tab.down; break;
    case UC_TABLE_HANN: v = &c->tab.hann; break;
    case UC_TABLE_H_UP: v = &c->tab.h_up; break;
    case UC_TABLE_H_DOWN: v = &c->tab.h_down; break;
    case UC_TABLE_CARRIER_C: v = &c->tab.carrier_c; break;
    case UC_TABLE_CARRIER_S: v = &c->tab.carrier_s; break;
    case UC_TABLE_FIR: v = &c->tab.fir; break;
    case UC_TABLE_TEMPLATE: v = &c->stab.tmpl; break;
    default: return fail(-EINVAL, "uc_get_table: unknown table %d", id);
  }
  if (v->empty()) return fail(-ENOENT, "uc_get_table: table %d does not exist for this variant", id);
  if (cap < v->size()) return fail(-ENOSPC, "uc_get_table: need %zu floats", v->size());
  memcpy(out, v->data(), v->size() * sizeof(float));
  return (int)v->size();
}

int32_t uc_idx2freq(const uc_ctx* c, uint32_t idx) {
  if (!c) return 0;
  const uint32_t n = c->cfg.n;
  if (c->cfg.variant == UC_IQ && !iq_baseband(c))  // experiments/iq_modulation/Src/main.c:112-114
    return (int32_t)(uint32_t)(c->cfg.fs * (float)idx / (float)n);
  const uint32_t ifs = (uint32_t)(int32_t)c->cfg.fs;
  if (idx < n / 2) return (int32_t)(ifs * idx / n);
  return (int32_t)((ifs * (n - idx) / n) * 0xFFFFFFFFu);
}

// The counter of one dynamically dealt launch.  The kernels leave a counter at zero when their last workgroup exits
// (uc_dev.hpp: handout_leave), so a slot is zero whenever no launch is using it and nothing is written here.
//   eager launch : the next slot of the context's ring; *slot = its index (pass it to work_counter_launched() behind
//                  the launch).  If the launch that last used that slot is still running (64 or more launches of ONE
//                  context in flight on several streams) the counter would be shared: *out = nullptr, the caller deals
//                  this launch statically.  While the context has only ever launched on one stream, stream order is
//                  the guard and no event is recorded or queried.
//   capture      : a slot the graph owns from now on (kGraphSlots per context, never recycled): two graphs replayed on
//                  two streams never share a counter, and a graph's own replays are serialised by the runtime.
//                  *slot = -1.  When the graph slots are used up: nullptr (static deal).
// Event queries are "potentially unsafe" calls: while ANY stream of the thread is being captured in the global capture mode
// (torch.cuda.graph's default) they are refused AND invalidate that capture.  The guard's events have nothing to do with
// a capture in progress, so its calls run with the thread's capture mode switched to relaxed for their duration (what
// allocators that must touch the runtime during someone else's capture do).
struct RelaxedCapture {
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  bool ok;
  RelaxedCapture() { ok = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess; if (!ok) (void)hipGetLastError(); }
  ~RelaxedCapture() { if (ok) (void)hipThreadExchangeStreamCaptureMode(&mode); }
};

static int take_work_counter(uc_ctx* c, hipStream_t stream, unsigned int** out, int* slot) {
  *out = nullptr;
  *slot = -1;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  unsigned idx;
  if (capturing) {
    if (c->graph_next >= kGraphSlots) {
      if (!c->graph_slots_warned) {  // once per context: the launch still works, dealt statically (a few percent slower)
        c->graph_slots_warned = true;
        fprintf(stderr, "libuchirp: context %p has handed out all %u graph-owned hand-out counters (one per captured launch, "
                        "never recycled); further captured launches use the static deal\n", (void*)c, kGraphSlots);
      }
      return 0;
    }
    idx = kWorkSlots + c->graph_next++;
  } else {
    // The guard below records and queries events (under RelaxedCapture, above).  An error from any of them must never fail
    // the launch: give the counter up and deal this launch statically (same results).  No call here ever waits for the
    // device.
    auto give_up = [&](hipError_t) {
      (void)hipGetLastError();
      return 0;
    };
    const RelaxedCapture relaxed;
    if (!c->multi_stream) {
      if (!c->ring_stream_set) {
        c->ring_stream = stream;
        c->ring_stream_set = true;
      } else if (stream != c->ring_stream) {
        hipStreamCaptureStatus rcap = hipStreamCaptureStatusNone;
        const bool ring_capturing = c->ring_stream && hipStreamIsCapturing(c->ring_stream, &rcap) == hipSuccess &&
                                    rcap != hipStreamCaptureStatusNone;
        if (ring_capturing) return give_up(hipSuccess);  // (an event recorded there would become a graph node)
        const hipError_t e = hipEventRecord(c->switch_ev, c->ring_stream);
        if (e != hipSuccess) {
          // the first stream no longer exists, or a capture elsewhere forbids the call: nothing is known about the slots
          // used so far -- retire the ring for good (every slot stays "in use before the switch" until an event says
          // otherwise, which none will: static deal for this context's eager launches from here on)
          (void)hipGetLastError();
          for (unsigned i = 0; i < kWorkSlots; i++) c->wait_switch[i] = c->slot_used[i];
          c->switch_lost = true;
        } else {
          for (unsigned i = 0; i < kWorkSlots; i++) c->wait_switch[i] = c->slot_used[i];
        }
        c->multi_stream = true;
      }
    }
    idx = c->work_next % kWorkSlots;
    if (c->multi_stream) {
      if (c->wait_switch[idx]) {
        if (c->switch_lost) return 0;
        const hipError_t q = hipEventQuery(c->switch_ev);
        if (q == hipErrorNotReady) return 0;  // launches from before the switch still run: deal this one statically
        if (q != hipSuccess) return give_up(q);
        for (unsigned i = 0; i < kWorkSlots; i++) c->wait_switch[i] = false;
      }
      if (c->work_busy[idx]) {
        const hipError_t q = hipEventQuery(c->work_ev[idx]);
        if (q == hipErrorNotReady) return 0;  // still in flight: do not advance, deal this launch statically
        if (q != hipSuccess) return give_up(q);
        c->work_busy[idx] = false;
      }
    }
    c->work_next++;
    *slot = (int)idx;
  }
  *out = (unsigned int*)((char*)c->d_work + (size_t)idx * kWorkStride);
  return 0;
}

// behind the launch that uses ring slot `slot` (no-op for -1: static deal or a graph-owned slot)
static int work_counter_launched(uc_ctx* c, hipStream_t stream, int slot) {
  if (slot < 0) return 0;
  c->slot_used[slot] = true;
  if (!c->multi_stream) return 0;  // one stream so far: stream order is the guard
  const RelaxedCapture relaxed;
  const hipError_t e = hipEventRecord(c->work_ev[slot], stream);
  if (e != hipSuccess) {
    // (a capture on another stream forbids the call): the launch is out and correct; without its event the slot cannot be
    // shown free again, so it stays busy -- later launches that land on it are dealt statically
    (void)hipGetLastError();
    c->wait_switch[slot] = true;
    c->switch_lost = true;
    return 0;
  }
  c->work_busy[slot] = true;
  return 0;
}

// uc_clock_probe: where the stamps of the launch about to be made go (nullptr when the probe is off): `waves` x 4 words,
// zeroed on the launch stream in front of the kernel (a wave that leaves before the loop writes nothing)
static int clock_buffer(uc_ctx* c, size_t grid, int waves_per_wg, hipStream_t stream, unsigned long long** out) {
  *out = nullptr;
  if (!c->clock_probe) return 0;
  const size_t waves = grid * (size_t)waves_per_wg;
  const int rc = c->s_clock.ensure(waves * 4 * sizeof(unsigned long long));
  if (rc) return rc;
  const hipError_t e = hipMemsetAsync(c->s_clock.p, 0, waves * 4 * sizeof(unsigned long long), stream);
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(clock stamps)");
  c->clock_waves = waves;
  *out = (unsigned long long*)c->s_clock.p;
  return 0;
}

// The launch of the band kernel (RX_REAL, SYNC_CPLX, DECHIRP_DOWN): the caller has filled in where the frames are and
// which outputs it wants (p.frames / n_frames / stride -- or the ROWS fields -- mag_mean, symbols, stats, magmax, spectrum,
// device pointers all); tables, window geometry, grid, group size and the hand-out counter are decided here.
static int band_launch(uc_ctx* c, uc::BandParams& p, int dtype, hipStream_t stream) {
  const int variant = c->cfg.variant;
  const size_t n_frames = p.n_frames;
  p.tab0 = c->d_tab0;
  p.tab1 = c->d_tab1;
  p.tw = c->d_tw;
  p.wide = c->tab.bandwidth2 > (uint32_t)uc::kBandNarrowMax ? 1u : 0u;
  p.mag_mean_scalar = c->cfg.mag_mean;
  p.snr_threshold = c->cfg.snr_threshold;
  p.bw2 = c->tab.bandwidth2;
  p.ifs = (uint32_t)(int32_t)c->cfg.fs;
  p.true_dc = (c->cfg.flags & UC_FLAG_TRUE_DC) ? 1u : 0u;
  p.debug = nullptr;
#if defined(UC_STAMPS)
  // diagnostic build only (libuchirp_stamps.so): where the per-phase stamps go
  if (const char* d = getenv("UC_DEBUG_PTR")) p.debug = (unsigned long long*)strtoull(d, nullptr, 0);
#endif
  const int mode = (variant == UC_SYNC_CPLX) ? uc::kModeCplx
                   : (variant == UC_DECHIRP_DOWN) ? uc::kModePair : uc::kModeRxReal;
  const bool rows = p.row_blocks != 0;
  // SYNC_CPLX runs two transforms per frame off two complex tables: at 2 waves/SIMD both tables stay in registers (at 3
  // the second one is loaded inside the loop, behind the frame prefetch in the in-order vector-memory queue):
  // 2.59e8 against 2.45e8 frames/s (profiles/r03_sync_cplx_waves.txt)
  // (the ROWS build exists at each mode's default occupancy, its WIDE form at 2 waves/SIMD: the value names the instantiation
  // that is dispatched -- uc_band_kernel.hip: UC_DISPATCH)
  const int waves = rows ? ((mode == uc::kModeCplx || p.wide) ? 2 : 3)
                         : (mode == uc::kModeCplx && !c->band_waves_set) ? 2 : c->band_waves;
  // uc_window_spectrum runs the SAME two-round build as the statistics path when the windows fit it (bandwidth2 <= 191): the
  // instantiation that also stores the window bins (uc_band_kernel.hip: SPEC), so that what the device captures are
  // compared with is the arithmetic of the throughput kernel
  const bool spec = p.spectrum != nullptr && !p.wide;
  // frames that overlap (stride < n) run the default build with default-policy loads -- a kernel of its own, asked for its own
  // occupancy (the dispatch takes it for RX_REAL at 3 and SYNC_CPLX at 2 waves/SIMD only)
  const bool overlap = !rows && !spec && !p.wide && p.stride < (size_t)uc::kN && mode != uc::kModePair &&
                       ((mode == uc::kModeRxReal && waves == 3) || (mode == uc::kModeCplx && waves == 2));
  int& bpc = c->band_blocks_per_cu[rows ? (p.wide ? 4 : 3) : (p.wide ? 1 : (spec ? 2 : (overlap ? 5 : 0)))][mode][dtype == UC_DTYPE_I32 ? 0 : 1];
  if (bpc == 0) bpc = uc::band_max_blocks_per_cu(mode, dtype, waves, p.wide != 0, spec, rows, overlap);
  size_t grid = (size_t)c->num_cu * (size_t)bpc;
  // DECHIRP_DOWN (frame pairs, the HBM-bound one) runs at the loads-only floor of this kernel structure, and that floor is
  // lower with fewer concurrent streams: 5 workgroups per CU instead of the 6 that fit: 7.69 against 7.54e8 frames/s,
  // 4 per CU 7.57, 3 per CU 6.86 (profiles/r03_band_knock.txt)
  if (mode == uc::kModePair && !p.wide && bpc > 5) grid = (size_t)c->num_cu * 5;
  if (c->grid_override > 0) grid = (size_t)c->grid_override;
  // units of work: frames, or frame pairs (DECHIRP_DOWN).  Groups of `band_group` units; smaller ones when the batch
  // would not give every workgroup a few (a small batch then still spreads over the whole chip)
  p.unpaired = (mode == uc::kModePair && (c->cfg.flags & UC_FLAG_NO_FRAME_PAIRS)) ? 1u : 0u;
  const size_t units = (mode == uc::kModePair && !p.unpaired) ? (n_frames + 1) / 2 : n_frames;
  uint32_t group = (uint32_t)c->band_group;
  if (waves >= 4 && group > 32) group = 32;  // (the ring of the 4-waves-per-SIMD build holds 32 frames)
  if (rows && group > 32) group = 32;        // (the ROWS build describes a group's units by ONE 32-bit word)
  const uint32_t group_cap = group;
  // (one-block calls of a live state, p.need: groups of whole rows -- the walk reads a group's need words as bytes of one word)
  const uint32_t group_min = (rows && p.need) ? 8u : 1u;
  if (group < group_min) group = group_min;
  while (group > group_min && units < (size_t)group * grid * 4) group >>= 1;
  const size_t ngroups = (units + group - 1) / group;
  if (grid > ngroups) grid = ngroups;
  p.group_log2 = 0;
  while ((1u << p.group_log2) < group) p.group_log2++;
  p.work_ctr = nullptr;
  int wslot = -1;
  // Dynamic hand-out only for batches big enough to keep full groups: a launch of a few dozen frames per workgroup is over
  // before the skew between workgroups that the tickets even out has built up, and pays for them -- 32 768 frames (the new
  // FIFO offsets of 4096 live streams): 0.101 ms dealt statically, 0.166 ms with tickets; 131 072: 0.309 / 0.322; from
  // 524 288 on the same (profiles/r05_live_deal.txt)
  // ... and not for the masked steps of live receivers (p.need): the walk of the ROWS build fetches the need words of the group
  // that FOLLOWS while it works on a group, which it can only do when it knows which group that is (g + gridDim.x); a masked
  // group lasts 20-40 us and 16 384 tickets on one word were felt (r5: 65 536 idle RX_REAL streams 0.663 -> 0.627 ms dealt
  // statically, profiles/r05_live_idle.txt)
  const bool masked = rows && p.need != nullptr;
  if (!c->static_deal && !masked && group >= 2 && group == group_cap && ngroups > grid) {
    const int wrc = take_work_counter(c, stream, &p.work_ctr, &wslot);  // dynamic hand-out
    if (wrc) return wrc;
  }
  if (c->clock_probe)
    if (int crc = clock_buffer(c, grid, 2, stream, &p.debug)) return crc;
  int lrc = (c->clock_probe ? uc::clk::launch_band : uc::launch_band)(mode, dtype, waves, p, (int)grid, stream);
  if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "band kernel launch");
  return work_counter_launched(c, stream, wslot);
}

static int process_batch_impl(uc_ctx* c, const void* frames, int dtype, size_t n_frames, size_t stride_elems,
                              const float* mag_mean, uint8_t* symbols, uc_stats* stats, float2* d_magmax, void* hip_stream,
                              bool mapped = false, float* d_spectrum = nullptr);

int uc_process_batch(uc_ctx* c, const void* frames, int dtype, size_t n_frames, size_t stride_elems,
                     const float* mag_mean, uint8_t* symbols, uc_stats* stats, void* hip_stream) {
  return process_batch_impl(c, frames, dtype, n_frames, stride_elems, mag_mean, symbols, stats, nullptr, hip_stream);
}

// d_magmax: device, (up, down) mag_max per frame, nullable (internal: uc_receive_stream)
// mapped  : every pointer is device-accessible as it stands (internal: the pinned, mapped frame slot of uc_process_frame)
// d_spectrum: device, the window bins of every frame (internal: uc_window_spectrum; band variants only)
static int process_batch_impl(uc_ctx* c, const void* frames, int dtype, size_t n_frames, size_t stride_elems,
                              const float* mag_mean, uint8_t* symbols, uc_stats* stats, float2* d_magmax, void* hip_stream,
                              bool mapped, float* d_spectrum) {
  if (!c) return fail(-EINVAL, "uc_process_batch: NULL ctx");
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32)
    return fail(-EINVAL, "uc_process_batch: dtype %d is neither UC_DTYPE_I32 nor UC_DTYPE_F32", dtype);
  if (c->cfg.variant == UC_STREAM)
    return fail(-EINVAL, "uc_process_batch: UC_STREAM has no frames, use uc_process_stream");
  if (n_frames == 0) return 0;
  if (!frames) return fail(-EINVAL, "uc_process_batch: frames is NULL");
  if (n_frames >= ((size_t)1 << 31)) return fail(-EINVAL, "uc_process_batch: at most 2^31 - 1 frames per call");
  const uint32_t n = c->cfg.n;
  if (stride_elems == 0) stride_elems = n;
  const int variant = c->cfg.variant;
  const int spf = uc_stats_per_frame(c);
  const int halo = uc_iq_halo(c);

  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;

  const size_t span = (n_frames - 1) * stride_elems + n + (size_t)halo;  // elements touched
  bool any_host_out = false;

  const void* d_frames = frames;
  if (!mapped && !is_device_ptr(frames)) {
    int rc = c->s_frames.ensure(span * 4);
    if (rc) return rc;
    const char* src = (const char*)frames - (size_t)halo * 4;
    e = hipMemcpyAsync(c->s_frames.p, src, span * 4, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(frames)");
    d_frames = (const char*)c->s_frames.p + (size_t)halo * 4;
  }
  const float* d_mm = mag_mean;
  if (mag_mean && !mapped && !is_device_ptr(mag_mean)) {
    int rc = c->s_mm.ensure(n_frames * 2 * sizeof(float));
    if (rc) return rc;
    e = hipMemcpyAsync(c->s_mm.p, mag_mean, n_frames * 2 * sizeof(float), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(mag_mean)");
    d_mm = (const float*)c->s_mm.p;
  }
  uint8_t* d_sym = symbols;
  if (symbols && !mapped && !is_device_ptr(symbols)) {
    int rc = c->s_sym.ensure(n_frames);
    if (rc) return rc;
    d_sym = (uint8_t*)c->s_sym.p;
    any_host_out = true;
  }
  uc_stats* d_stats = stats;
  if (stats && !mapped && !is_device_ptr(stats)) {
    int rc = c->s_stats.ensure(n_frames * (size_t)spf * sizeof(uc_stats));
    if (rc) return rc;
    d_stats = (uc_stats*)c->s_stats.p;
    any_host_out = true;
  }

  if (variant == UC_IQ) {
    uc::IqParams ip;
    memset(&ip, 0, sizeof(ip));
    ip.frames = d_frames;
    ip.n_frames = n_frames;
    ip.stride = stride_elems;
    ip.carrier = c->d_tab0;
    ip.chirp_hann = c->d_tab1;
    ip.tw = c->d_tw;
    ip.mag_mean = d_mm;
    ip.symbols = d_sym;
    ip.stats = d_stats;
    for (int k = 0; k < uc::kFirTapsDev; k++) ip.fir[k] = c->tab.fir[k];
    ip.mag_mean_scalar = c->cfg.mag_mean;
    ip.fs = c->cfg.fs;
    ip.idx_left_zero = c->tab.idx_left_zero;
    ip.center = c->tab.center;
    ip.bw2 = c->tab.bandwidth2;
    ip.bw4 = c->tab.bandwidth4;
    const bool bb = iq_baseband(c);
    if (bb) {
      // the windows straddle DC: the kernel walks UNWRAPPED bins n - bandwidth ... n + bandwidth (taken mod n)
      ip.chirp_hann2 = c->d_tab2;
      ip.baseband = 1u;
      ip.center = n;
      ip.ifs = (uint32_t)(int32_t)c->cfg.fs;
      ip.snr_threshold = c->cfg.snr_threshold;
    }
    ip.fir_mfma = (n == 1024 && c->iq_fir_mfma) ? c->d_aux : nullptr;
    ip.stagger = c->iq_stagger;
    int& iq_bpc = c->iq_blocks_per_cu[dtype == UC_DTYPE_I32 ? 0 : 1];
    if (iq_bpc == 0) iq_bpc = uc::iq_max_blocks_per_cu(dtype, (int)n, bb ? 1 : 0, ip.fir_mfma ? 1 : 0,
                                                      ip.bw2 <= (n == 1024 ? 32u : 64u) ? 1 : 0);
    size_t grid = (size_t)c->num_cu * (size_t)iq_bpc;
    if (c->grid_override > 0) grid = (size_t)c->grid_override;
    if (grid > n_frames) grid = n_frames;
    // groups of up to 64 consecutive frames (one finaliser drain each), dealt round robin;
    // smaller groups when the batch would not give every workgroup one
    ip.group = (uint32_t)c->iq_group;
    if (bb && n == 2048 && ip.group > 32) ip.group = 32;  // (the base-band ring of the n = 2048 kernel holds 32 frames)
    while (ip.group > 1 && n_frames < (size_t)ip.group * grid) ip.group >>= 1;
    {
      const size_t ngroups = (n_frames + ip.group - 1) / ip.group;
      if (grid > ngroups) grid = ngroups;
    }
    ip.work_ctr = nullptr;
    int wslot = -1;
    // (tickets only for batches of at least four full groups per workgroup: below that the launch is over before the skew they
    // even out has built up, and the tickets cost more than they save -- 65 536 base-band frames: +13 % dealt statically,
    // 262 144: -3 %; the band kernel's rule, profiles/r05_live_deal.txt)
    if (!c->static_deal && ip.group >= 2 && n_frames >= (size_t)4 * (size_t)c->iq_group * grid) {
      const size_t ngroups = (n_frames + ip.group - 1) / ip.group;
      if (ngroups > grid) {  // dynamic hand-out
        const int wrc = take_work_counter(c, stream, &ip.work_ctr, &wslot);
        if (wrc) return wrc;
      }
    }
    if (int crc = clock_buffer(c, grid, n == 1024 ? 1 : 2, stream, &ip.debug)) return crc;
    int lrc = (c->clock_probe ? uc::clk::launch_iq : uc::launch_iq)(dtype, ip, (int)grid, stream, (int)n);
    if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "iq kernel launch");
    if (int erc = work_counter_launched(c, stream, wslot)) return erc;
    goto copy_back;
  }
  if (variant == UC_COMPRESS) {
    uc::FullParams fp;
    memset(&fp, 0, sizeof(fp));
    fp.frames = d_frames;
    fp.n_frames = n_frames;
    fp.stride = stride_elems;
    fp.hann = (const float*)c->d_tab1;
    fp.hn = c->d_tab0;
    fp.tw = c->d_tw;
    fp.mag_mean = d_mm;
    fp.symbols = d_sym;
    fp.stats = d_stats;
    fp.mag_mean_scalar = c->cfg.mag_mean;
    int& full_bpc = c->full_blocks_per_cu[dtype == UC_DTYPE_I32 ? 0 : 1];
    if (full_bpc == 0) full_bpc = uc::compress_max_blocks_per_cu(dtype);
    size_t grid = (size_t)c->num_cu * (size_t)full_bpc;
    if (c->grid_override > 0) grid = (size_t)c->grid_override;
    fp.unpaired = (c->cfg.flags & UC_FLAG_NO_FRAME_PAIRS) ? 1u : 0u;
    const size_t npairs = fp.unpaired ? n_frames : (n_frames + 1) / 2;  // units of work
    if (grid > npairs) grid = npairs;
    fp.work_ctr = nullptr;
    fp.chunk_log2 = 0;
    int wslot = -1;
    // (at least four chunks per workgroup: 16 384 pairs +22 % dealt statically, 65 536 pairs -2 %)
    if (!c->static_deal && c->compress_chunk >= 2 && npairs >= (size_t)4 * (size_t)c->compress_chunk * grid) {
      const int wrc = take_work_counter(c, stream, &fp.work_ctr, &wslot);  // dynamic hand-out of chunks of consecutive pairs
      if (wrc) return wrc;
      if (fp.work_ctr) {
        while ((1u << fp.chunk_log2) < (unsigned)c->compress_chunk) fp.chunk_log2++;
        const size_t nchunks = (npairs + ((size_t)1 << fp.chunk_log2) - 1) >> fp.chunk_log2;
        if (grid > nchunks) grid = nchunks;
      }
    }
    if (int crc = clock_buffer(c, grid, 2, stream, &fp.debug)) return crc;
    int lrc = (c->clock_probe ? uc::clk::launch_compress : uc::launch_compress)(dtype, fp, (int)grid, stream);
    if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "compress kernel launch");
    if (int erc = work_counter_launched(c, stream, wslot)) return erc;
    goto copy_back;
  }
  {
    uc::BandParams p;
    memset(&p, 0, sizeof(p));
    p.frames = d_frames;
    p.n_frames = n_frames;
    p.stride = stride_elems;
    p.mag_mean = d_mm;
    p.symbols = d_sym;
    p.stats = d_stats;
    p.magmax = d_magmax;
    p.spectrum = d_spectrum;
    if (int brc = band_launch(c, p, dtype, stream)) return brc;
  }
copy_back:

  if (any_host_out) {
    if (symbols && d_sym != symbols) {
      e = hipMemcpyAsync(symbols, d_sym, n_frames, hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(symbols)");
    }
    if (stats && d_stats != stats) {
      e = hipMemcpyAsync(stats, d_stats, n_frames * (size_t)spf * sizeof(uc_stats), hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(stats)");
    }
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  return 0;
}

int uc_clock_probe(uc_ctx* c, int on) {
  if (!c) return fail(-EINVAL, "uc_clock_probe: NULL ctx");
  c->clock_probe = on != 0;
  c->clock_waves = 0;
  return 0;
}

int uc_clock_read(uc_ctx* c, uc_clock* out) {
  if (!c || !out) return fail(-EINVAL, "uc_clock_read: NULL argument");
  memset(out, 0, sizeof(*out));
  if (!c->clock_probe || c->clock_waves == 0) return fail(-ENODATA, "uc_clock_read: no launch since uc_clock_probe(ctx, 1)");
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) return hip_fail(e, "hipDeviceSynchronize");
  std::vector<unsigned long long> w(c->clock_waves * 4);
  e = hipMemcpy(w.data(), c->s_clock.p, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(clock stamps)");
  // per wave: [0] shader cycles of its loop (low 40 bits; the bits above name the CU), [1] the same span in ticks of the
  // constant 100 MHz clock, [2] / [3] absolute start / end ticks
  std::vector<double> ghz, cyc;
  unsigned long long t0 = ~0ull, t1 = 0;
  for (size_t i = 0; i < c->clock_waves; i++) {
    const unsigned long long cycles = w[4 * i] & 0xffffffffffull, ticks = w[4 * i + 1];
    if (ticks == 0) continue;  // a wave that had nothing to do
    ghz.push_back((double)cycles / (double)ticks * 0.1);
    cyc.push_back((double)cycles);
    if (w[4 * i + 2] < t0) t0 = w[4 * i + 2];
    if (w[4 * i + 3] > t1) t1 = w[4 * i + 3];
  }
  if (ghz.empty()) return fail(-ENODATA, "uc_clock_read: the last launch stamped no wave");
  std::sort(ghz.begin(), ghz.end());
  std::sort(cyc.begin(), cyc.end());
  out->shader_ghz = ghz[ghz.size() / 2];
  out->wave_cycles = cyc[cyc.size() / 2];
  out->span_us = (double)(t1 - t0) * 0.01;
  out->waves = (uint32_t)ghz.size();
  return 0;
}

int uc_clock_stamps(uc_ctx* c, uint64_t* words, size_t cap_words) {
  if (!c) return fail(-EINVAL, "uc_clock_stamps: NULL ctx");
  if (!c->clock_probe || c->clock_waves == 0) return fail(-ENODATA, "uc_clock_stamps: no launch since uc_clock_probe(ctx, 1)");
  const size_t nw = c->clock_waves * 4;
  if (!words || cap_words < nw) return (int)nw;  // (size query)
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(words, c->s_clock.p, nw * sizeof(uint64_t), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "uc_clock_stamps");
  return (int)nw;
}

int uc_set_table(uc_ctx* c, int table_id, const float* data, size_t count) {
  if (!c || !data) return fail(-EINVAL, "uc_set_table: NULL argument");
  const int v = c->cfg.variant;
  if (v != UC_RX_REAL && v != UC_SYNC_CPLX && v != UC_DECHIRP_DOWN)
    return fail(-ENOTSUP, "uc_set_table: variant %d derives further tables from its references (RX_REAL, SYNC_CPLX, "
                          "DECHIRP_DOWN only)", v);
  std::vector<float>* dst = nullptr;
  switch (table_id) {
    case UC_TABLE_UP: dst = &c->tab.up; break;
    case UC_TABLE_DOWN: dst = &c->tab.down; break;
    case UC_TABLE_HANN: dst = &c->tab.hann; break;
    default: return fail(-EINVAL, "uc_set_table: table %d cannot be replaced (UC_TABLE_UP, _DOWN, _HANN)", table_id);
  }
  if (dst->empty()) return fail(-ENOENT, "uc_set_table: table %d does not exist for this variant", table_id);
  if (count != dst->size()) return fail(-EINVAL, "uc_set_table: table %d holds %zu floats, got %zu", table_id, dst->size(), count);
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  e = hipDeviceSynchronize();  // no launch of this context may still be reading the old tables
  if (e != hipSuccess) return hip_fail(e, "hipDeviceSynchronize");
  // all or nothing: the host copy (what uc_get_table reports) changes only if every device table took the new reference;
  // on a failed upload the old reference is put back on both sides
  const std::vector<float> old = *dst;
  memcpy(dst->data(), data, count * sizeof(float));
  int rc = upload_device_tables(c);
  if (rc) {
    const std::string why = g_err;
    *dst = old;
    (void)upload_device_tables(c);  // (best effort: the same copies that just failed may fail again)
    g_err = why;
  }
  return rc;
}

// Diagnostic: hand-out counters that are not zero although no launch of the context is in flight (waits for the device).
// Always 0: every dynamically dealt launch leaves its counter at zero when its last workgroup exits (uc_dev.hpp:
// handout_leave).  A non-zero value means a kernel path returned without passing that exit -- the next launch on that slot
// would skip work groups silently.  tests/test_gpu_handout.py asserts it behind every kernel family.
int uc_debug_busy_counters(uc_ctx* c) {
  if (!c) return fail(-EINVAL, "uc_debug_busy_counters: NULL ctx");
  hipError_t e = hipSetDevice(c->device);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) return hip_fail(e, "hipDeviceSynchronize");
  const size_t words = (size_t)(kWorkSlots + kGraphSlots) * kWorkStride / sizeof(unsigned int);
  std::vector<unsigned int> w(words);
  e = hipMemcpy(w.data(), c->d_work, words * sizeof(unsigned int), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(work counters)");
  int busy = 0;
  for (unsigned int v : w) busy += v != 0;
  return busy;
}

int uc_window_bins(const uc_ctx* c) {
  if (!c) return fail(-EINVAL, "uc_window_bins: NULL ctx");
  const int v = c->cfg.variant;
  if (v != UC_RX_REAL && v != UC_SYNC_CPLX && v != UC_DECHIRP_DOWN)
    return fail(-ENOTSUP, "uc_window_bins: variant %d has no windows around DC", v);
  return (int)(2 * c->tab.bandwidth2 + 1);
}

int uc_window_spectrum(uc_ctx* c, const void* frames, int dtype, size_t n_frames, size_t stride_elems, float* mags,
                       void* hip_stream) {
  if (!c) return fail(-EINVAL, "uc_window_spectrum: NULL ctx");
  const int wb = uc_window_bins(c);
  if (wb < 0) return wb;
  if (n_frames == 0) return 0;
  if (!mags) return fail(-EINVAL, "uc_window_spectrum: mags is NULL");
  const size_t count = n_frames * (size_t)uc_stats_per_frame(c) * (size_t)wb;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;
  float* d_out = mags;
  const bool host_out = !is_device_ptr(mags);
  if (host_out) {
    const int rc = c->s_spec.ensure(count * sizeof(float));
    if (rc) return rc;
    d_out = (float*)c->s_spec.p;
  }
  const int rc = process_batch_impl(c, frames, dtype, n_frames, stride_elems, nullptr, nullptr, nullptr, nullptr, hip_stream,
                                    false, d_out);
  if (rc) return rc;
  if (host_out) {
    e = hipMemcpyAsync(mags, d_out, count * sizeof(float), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(window spectrum)");
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  return 0;
}

// the sinc^5 byte tables on the device and the kernel's LDS opt-in, once per context
static int sinc5_prepare(uc_ctx* c) {
  if (!c->d_cic4) {
    std::vector<int32_t> t4, t1;
    uc::build_sinc5_tables(t4, t1);
    int rc = upload((void**)&c->d_cic4, t4.data(), t4.size() * sizeof(int32_t));
    if (!rc) rc = upload((void**)&c->d_cic1, t1.data(), t1.size() * sizeof(int32_t));
    if (rc) return rc;
  }
  if (c->cic_blocks_per_cu == 0) {
    c->cic_blocks_per_cu = uc::sinc5_max_blocks_per_cu();
    if (c->cic_blocks_per_cu <= 0) {
      c->cic_blocks_per_cu = 0;
      return fail(-ENOMEM, "uc_dfsdm_sinc5: the kernel's LDS tables do not fit this device");
    }
  }
  return 0;
}

// uc_dfsdm_sinc5_streams on device buffers: n_words NEW words of every stream, history carried in d_hist ([n_streams][4])
// (update_hist false: d_hist is only read -- uc_dfsdm_sinc5, where it is the head of the caller's input)
static int sinc5_streams_launch(uc_ctx* c, const uint32_t* d_pdm, size_t n_streams, size_t n_words, size_t stride,
                                const uint32_t* d_hist, bool update_hist, int32_t* d_out, size_t out_stride,
                                hipStream_t stream) {
  if (n_streams == 0 || n_words == 0) return 0;
  if ((((uintptr_t)d_pdm | (uintptr_t)d_out | (uintptr_t)d_hist) & 15u) != 0 ||
      (n_streams > 1 && ((stride & 3u) != 0 || (out_stride & 3u) != 0)))
    return fail(-EINVAL, "uc_dfsdm_sinc5_streams: device buffers must be 16-byte aligned and the strides multiples of 4 words");
  if (int rc = sinc5_prepare(c)) return rc;
  uc::CicParams cp;
  memset(&cp, 0, sizeof(cp));
  cp.pdm = d_pdm;
  cp.n_words = n_words;
  cp.out = d_out;
  cp.t4 = c->d_cic4;
  cp.t1 = c->d_cic1;
  cp.n_streams = n_streams;
  cp.stride = stride;
  cp.out_stride = out_stride;
  cp.hist = d_hist;
  cp.update_hist = update_hist ? 1u : 0u;
  size_t grid = (size_t)c->num_cu * (size_t)c->cic_blocks_per_cu;
  if (c->grid_override > 0) grid = (size_t)c->grid_override;
  const size_t per_block = (size_t)uc::sinc5_waves_per_block();
  // Tiles of 256 words; one wave walks a SEGMENT of up to 8 tiles front to back (uc_cic_kernel.hip).  A live block (2048
  // words) is one segment.  Few streams: shorter segments, so that every wave of the grid has one.
  const size_t tiles = (n_words + 255) / 256;
  size_t nseg = (tiles + 7) / 8;
  const size_t spread = (grid * per_block + n_streams - 1) / n_streams;  // segments per stream that fill the grid
  if (nseg < spread) nseg = spread < tiles ? spread : tiles;
  const size_t tps = (tiles + nseg - 1) / nseg;
  nseg = (tiles + tps - 1) / tps;
  if (nseg * n_streams >= ((size_t)1 << 31)) return fail(-EINVAL, "uc_dfsdm_sinc5_streams: too many segments in one call");
  cp.tps = (uint32_t)tps;
  cp.nseg = (uint32_t)nseg;
  cp.units = (uint32_t)(nseg * n_streams);
  uc::rows_divisor(cp.nseg, &cp.div_magic, &cp.div_shift);
  const size_t need = ((size_t)cp.units + per_block - 1) / per_block;
  if (grid > need) grid = need;
  if (c->clock_probe) {
    if (c->clk_cic_blocks == 0) c->clk_cic_blocks = uc::clk::sinc5_max_blocks_per_cu();  // (the twin needs the same LDS opt-in)
    if (c->clk_cic_blocks <= 0) return fail(-ENOMEM, "uc_dfsdm_sinc5: the clock-stamped kernel's LDS tables do not fit");
    if (int crc = clock_buffer(c, grid, uc::clk::sinc5_waves_per_block(), stream, &cp.debug)) return crc;
  }
  const int lrc = (c->clock_probe ? uc::clk::launch_sinc5 : uc::launch_sinc5)(cp, (int)grid, stream);
  if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "sinc5 kernel launch");
  return 0;
}

int uc_dfsdm_sinc5_streams(uc_ctx* c, const uint32_t* pdm_words, size_t n_streams, size_t n_words, size_t stride_words,
                           uint32_t* history, int32_t* words_out, size_t out_stride_words, void* hip_stream) {
  if (!c) return fail(-EINVAL, "uc_dfsdm_sinc5_streams: NULL ctx");
  if (n_streams == 0 || n_words == 0) return 0;
  if (!pdm_words || !words_out || !history) return fail(-EINVAL, "uc_dfsdm_sinc5_streams: NULL buffer");
  if (stride_words == 0) stride_words = n_words;
  if (out_stride_words == 0) out_stride_words = n_words;
  if (stride_words < n_words || out_stride_words < n_words)
    return fail(-EINVAL, "uc_dfsdm_sinc5_streams: streams overlap (stride < %zu words)", n_words);
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;
  // host buffers are staged through the context (padded to whole 16-byte rows), device buffers are used where they lie
  const bool in_host = !is_device_ptr(pdm_words), hist_host = !is_device_ptr(history), out_host = !is_device_ptr(words_out);
  const uint32_t* d_in = pdm_words;
  size_t in_stride = stride_words;
  if (in_host) {
    in_stride = (n_words + 3) & ~(size_t)3;
    if (int rc = c->s_cic_in.ensure(n_streams * in_stride * 4)) return rc;
    e = hipMemcpy2DAsync(c->s_cic_in.p, in_stride * 4, pdm_words, stride_words * 4, n_words * 4, n_streams, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpy2DAsync(pdm)");
    d_in = (const uint32_t*)c->s_cic_in.p;
  }
  uint32_t* d_hist = history;
  if (hist_host) {
    if (int rc = c->s_cic_hist.ensure(n_streams * 16)) return rc;
    e = hipMemcpyAsync(c->s_cic_hist.p, history, n_streams * 16, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(history)");
    d_hist = (uint32_t*)c->s_cic_hist.p;
  }
  int32_t* d_out = words_out;
  size_t o_stride = out_stride_words;
  if (out_host) {
    o_stride = (n_words + 3) & ~(size_t)3;
    if (int rc = c->s_cic_out.ensure(n_streams * o_stride * 4)) return rc;
    d_out = (int32_t*)c->s_cic_out.p;
  }
  if (int rc = sinc5_streams_launch(c, d_in, n_streams, n_words, in_stride, d_hist, true, d_out, o_stride, stream)) return rc;
  if (out_host || hist_host) {
    if (out_host) {
      e = hipMemcpy2DAsync(words_out, out_stride_words * 4, d_out, o_stride * 4, n_words * 4, n_streams, hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpy2DAsync(words_out)");
    }
    if (hist_host) {
      e = hipMemcpyAsync(history, d_hist, n_streams * 16, hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(history)");
    }
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  return 0;
}

int uc_dfsdm_sinc5(uc_ctx* c, const uint32_t* pdm_words, size_t n_words, int32_t* words_out, void* hip_stream) {
  if (!c) return fail(-EINVAL, "uc_dfsdm_sinc5: NULL ctx");
  if (n_words <= 4) return 0;
  if (!pdm_words || !words_out) return fail(-EINVAL, "uc_dfsdm_sinc5: NULL buffer");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;
  const size_t n_out = n_words - 4;
  const uint32_t* d_in = pdm_words;
  if (!is_device_ptr(pdm_words)) {
    int rc = c->s_cic_in.ensure(n_words * 4);
    if (rc) return rc;
    e = hipMemcpyAsync(c->s_cic_in.p, pdm_words, n_words * 4, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(pdm)");
    d_in = (const uint32_t*)c->s_cic_in.p;
  } else if (((uintptr_t)pdm_words & 15u) != 0) {
    return fail(-EINVAL, "uc_dfsdm_sinc5: a device `pdm_words` pointer must be 16-byte aligned");
  }
  int32_t* d_out = words_out;
  const bool host_out = !is_device_ptr(words_out);
  if (host_out) {
    int rc = c->s_cic_out.ensure(n_out * 4);
    if (rc) return rc;
    d_out = (int32_t*)c->s_cic_out.p;
  } else if (((uintptr_t)words_out & 15u) != 0) {
    return fail(-EINVAL, "uc_dfsdm_sinc5: a device `words_out` pointer must be 16-byte aligned");
  }
  // one stream whose history lies in front of it: words 0 .. 3 are the history, words 4 .. the stream
  if (int rc = sinc5_streams_launch(c, d_in + 4, 1, n_out, n_out, d_in, false, d_out, n_out, stream)) return rc;
  if (host_out) {
    e = hipMemcpyAsync(words_out, d_out, n_out * 4, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(words_out)");
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  return 0;
}

int uc_stream_geometry(const uc_ctx* c, size_t n_samples, size_t* halo, size_t* n_out, size_t* n_blocks,
                       size_t* hop) {
  if (!c) return fail(-EINVAL, "uc_stream_geometry: NULL ctx");
  if (c->cfg.variant != UC_STREAM) return fail(-EINVAL, "uc_stream_geometry: the context is not UC_STREAM");
  const size_t h = c->stab.halo, hp = c->stab.hop, D = c->stab.decim;
  const size_t no = n_samples > h ? (n_samples - h) / D : 0;
  if (halo) *halo = h;
  if (n_out) *n_out = no;
  if (n_blocks) *n_blocks = (no + hp - 1) / hp;
  if (hop) *hop = hp;
  return 0;
}

int uc_stream_span(const uc_ctx* c, size_t n_samples, int world, int rank, size_t* first_sample, size_t* n_shard,
                   size_t* first_out, size_t* n_out) {
  if (!c) return fail(-EINVAL, "uc_stream_span: NULL ctx");
  if (c->cfg.variant != UC_STREAM) return fail(-EINVAL, "uc_stream_span: the context is not UC_STREAM");
  const size_t h = c->stab.halo, hp = c->stab.hop, D = c->stab.decim;
  const size_t no = n_samples > h ? (n_samples - h) / D : 0;
  const size_t nb = (no + hp - 1) / hp;
  size_t b0 = 0, bc = 0;
  const int rc = uc_partition(nb, world, rank, &b0, &bc);  // whole overlap-save blocks: boundaries as in the one-GPU run
  if (rc) return rc;
  size_t q0 = b0 * hp, q1 = (b0 + bc) * hp;
  if (q0 > no) q0 = no;
  if (q1 > no) q1 = no;
  const bool empty = q1 <= q0;
  if (first_sample) *first_sample = empty ? 0 : q0 * D;
  if (n_shard) *n_shard = empty ? 0 : h + (q1 - q0) * D;
  if (first_out) *first_out = q0;
  if (n_out) *n_out = empty ? 0 : q1 - q0;
  return 0;
}

int uc_process_stream(uc_ctx* c, const void* samples, int dtype, size_t n_samples, float* compressed,
                      uc_peak* peaks, void* hip_stream) {
  if (!c) return fail(-EINVAL, "uc_process_stream: NULL ctx");
  if (c->cfg.variant != UC_STREAM) return fail(-EINVAL, "uc_process_stream: the context is not UC_STREAM");
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32)
    return fail(-EINVAL, "uc_process_stream: dtype %d is neither UC_DTYPE_I32 nor UC_DTYPE_F32", dtype);
  size_t n_out = 0, n_blocks = 0;
  uc_stream_geometry(c, n_samples, nullptr, &n_out, &n_blocks, nullptr);
  if (n_out == 0) return 0;
  if (!samples) return fail(-EINVAL, "uc_process_stream: samples is NULL");

  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;

  const void* d_samples = samples;
  if (!is_device_ptr(samples)) {
    int rc = c->s_frames.ensure(n_samples * 4);
    if (rc) return rc;
    e = hipMemcpyAsync(c->s_frames.p, samples, n_samples * 4, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(samples)");
    d_samples = c->s_frames.p;
  } else if (((uintptr_t)samples & 15u) != 0) {
    return fail(-EINVAL, "uc_process_stream: a device `samples` pointer must be 16-byte aligned");
  }
  bool any_host_out = false;
  float* d_comp = compressed;
  if (compressed && !is_device_ptr(compressed)) {
    int rc = c->s_comp.ensure(n_out * sizeof(float));
    if (rc) return rc;
    d_comp = (float*)c->s_comp.p;
    any_host_out = true;
  }
  uc_peak* d_peaks = peaks;
  if (peaks && !is_device_ptr(peaks)) {
    int rc = c->s_peaks.ensure(n_blocks * sizeof(uc_peak));
    if (rc) return rc;
    d_peaks = (uc_peak*)c->s_peaks.p;
    any_host_out = true;
  }

  uc::StreamParams sp;
  memset(&sp, 0, sizeof(sp));
  sp.samples = d_samples;
  sp.n_samples = n_samples;
  sp.n_out = n_out;
  sp.n_blocks = n_blocks;
  sp.hn = c->d_tab0;
  sp.rot = c->d_tab1;
  sp.tw = c->d_tw;
  sp.compressed = d_comp;
  sp.peaks = d_peaks;
  for (int k = 0; k < 2 * uc::kFirTapsDev; k++) sp.ctap[k] = c->stab.ctap[k];
  const int D = (int)c->stab.decim;
  for (int sub = 0; sub < D / 2; sub++) {
    sp.rots[2 * sub] = c->stab.rot[2 * (size_t)(sub * (4096 / D))];
    sp.rots[2 * sub + 1] = c->stab.rot[2 * (size_t)(sub * (4096 / D)) + 1];
  }
  int& st_bpc = c->stream_blocks_per_cu[dtype == UC_DTYPE_I32 ? 0 : 1];
  if (st_bpc == 0) st_bpc = uc::stream_max_blocks_per_cu(dtype, D);
  size_t grid = (size_t)c->num_cu * (size_t)st_bpc;
  if (c->grid_override > 0) grid = (size_t)c->grid_override;
  if (grid > n_blocks) grid = n_blocks;
  if (n_blocks >= ((size_t)1 << 32)) return fail(-EINVAL, "uc_process_stream: at most 2^32 - 1 blocks per call");
  sp.work_ctr = nullptr;
  sp.chunk_log2 = 0;
  int wslot = -1;
  // (tickets only from sixteen chunks per workgroup on: 2^26 samples +50 % dealt statically, 2^28 +9 %, 2^31 -11 %)
  if (!c->static_deal && n_blocks >= (size_t)16 * (size_t)c->stream_chunk * grid) {
    // dynamic hand-out of chunks of consecutive blocks
    const int wrc = take_work_counter(c, stream, &sp.work_ctr, &wslot);
    if (wrc) return wrc;
    if (sp.work_ctr) {
      while ((1u << sp.chunk_log2) < (unsigned)c->stream_chunk) sp.chunk_log2++;
      const size_t nchunks = (n_blocks + ((size_t)1 << sp.chunk_log2) - 1) >> sp.chunk_log2;
      if (grid > nchunks) grid = nchunks;
    }
  }
  if (int crc = clock_buffer(c, grid, 2, stream, &sp.debug)) return crc;
  int lrc = (c->clock_probe ? uc::clk::launch_stream : uc::launch_stream)(dtype, D, sp, (int)grid, stream);
  if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "stream kernel launch");
  if (int erc = work_counter_launched(c, stream, wslot)) return erc;

  if (any_host_out) {
    if (compressed && d_comp != compressed) {
      e = hipMemcpyAsync(compressed, d_comp, n_out * sizeof(float), hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(compressed)");
    }
    if (peaks && d_peaks != peaks) {
      e = hipMemcpyAsync(peaks, d_peaks, n_blocks * sizeof(uc_peak), hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(peaks)");
    }
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  }
  return 0;
}

int uc_process_frame(uc_ctx* c, const int32_t* pcm_in, float mag_mean, uint8_t* symbol_out, uc_stats st[2]) {
  if (!c || !pcm_in) return fail(-EINVAL, "uc_process_frame: NULL argument");
  if (uc_iq_halo(c)) return fail(-EINVAL, "uc_process_frame: UC_IQ needs FIR history, use uc_process_batch");
  if (c->cfg.variant == UC_STREAM) return fail(-EINVAL, "uc_process_frame: UC_STREAM has no frames, use uc_process_stream");
  // One frame per call is the firmware's own granularity (dsp(), receiver/Src/main.c:183-231): no staging copies.
  // The frame, the noise floors, the histories and the symbol live in ONE pinned host slot that the GPU reads and
  // writes in place over PCIe (8 KiB in, 65 B out); the call is the host memcpy into the slot, one launch, one wait.
  const size_t n = c->cfg.n;
  const size_t off_mm = n * 4, off_st = off_mm + 64, off_sym = off_st + 2 * sizeof(uc_stats);
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  if (!c->h_slot) {
    e = hipHostMalloc(&c->h_slot, off_sym + 64, hipHostMallocMapped);
    if (e != hipSuccess) { c->h_slot = nullptr; return hip_fail(e, "hipHostMalloc(frame slot)"); }
  }
  char* slot = (char*)c->h_slot;
  memcpy(slot, pcm_in, n * 4);
  float* mm = (float*)(slot + off_mm);
  mm[0] = mm[1] = mag_mean;
  uc_stats* hs = (uc_stats*)(slot + off_st);
  uint8_t* hsym = (uint8_t*)(slot + off_sym);
  memset(hs, 0, 2 * sizeof(uc_stats));
  *hsym = UC_SYM_NONE;
  int rc = process_batch_impl(c, slot, UC_DTYPE_I32, 1, n, mm, hsym, hs, nullptr, nullptr, /*mapped=*/true);
  if (rc) return rc;
  e = hipStreamSynchronize(nullptr);
  if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize");
  if (symbol_out) *symbol_out = *hsym;
  if (st) memcpy(st, hs, sizeof(uc_stats) * (size_t)uc_stats_per_frame(c));
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// uc_receive_stream: the receiver's main loop over a recorded stream.
// DSP: ONE batched launch over every 256-sample offset of the zero-prefixed
// stream; control: main()'s switch (include/uchirp_mainloop.hpp, shared with the
// C++ host layer) replayed on the host over the (up, down) mag_max of every frame.
// ---------------------------------------------------------------------------
using uc::RxReplay;

extern "C" int uc_receive_stream_isr(uc_ctx* c, const void* samples, int dtype, size_t n_samples, const uint8_t* busy,
                                     char* text, size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace) {
  if (!c || !text || text_cap == 0) return fail(-EINVAL, "uc_receive_stream: NULL argument");
  text[0] = '\0';
  if (n_trace) *n_trace = 0;
  if (c->cfg.variant != UC_RX_REAL && c->cfg.variant != UC_SYNC_CPLX)
    return fail(-ENOTSUP, "uc_receive_stream: variant %d has no up/down state machine", (int)c->cfg.variant);
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32) return fail(-EINVAL, "uc_receive_stream: bad dtype %d", dtype);
  const uint32_t n = c->cfg.n;
  const size_t n_blocks = n_samples / n;
  if (n_blocks == 0) return 0;
  if (!samples) return fail(-EINVAL, "uc_receive_stream: samples is NULL");

  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  // The ISR (main.c:659-668) appends a block only when the main loop has consumed the previous one: a block that
  // arrives while `new_pcm_data` is still set is DROPPED, the FIFO is not shifted.  `busy[b] != 0` says the consumer
  // was still busy when block b arrived (NULL: never -- the GPU evaluates all of a block's dsp() calls at once).
  // The FIFO therefore only ever holds ACCEPTED blocks, in order: drop the others before the batched launch.
  std::vector<uint32_t> accepted;
  accepted.reserve(n_blocks);
  for (size_t b = 0; b < n_blocks; b++)
    if (!busy || !busy[b]) accepted.push_back((uint32_t)b);
  const size_t na = accepted.size();
  if (na == 0) return 0;
  // zero-prefixed copy of the accepted stream on the device: fifo_queue starts as 3n zeros (main.c:94)
  const size_t padded = (2 + na) * (size_t)n;
  const size_t n_frames = (padded - n) / 256 + 1;
  int rc = c->s_rx_pad.ensure(padded * 4);
  if (!rc) rc = c->s_rx_mag.ensure(n_frames * sizeof(float2));
  if (rc) return rc;
  char* d_pad = (char*)c->s_rx_pad.p;
  const hipMemcpyKind kind = is_device_ptr(samples) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  e = hipMemsetAsync(d_pad, 0, 2 * (size_t)n * 4, nullptr);
  for (size_t i = 0; e == hipSuccess && i < na;) {  // runs of consecutive accepted blocks: one copy each
    size_t jn = i + 1;
    while (jn < na && accepted[jn] == accepted[jn - 1] + 1) jn++;
    e = hipMemcpyAsync(d_pad + (2 + i) * (size_t)n * 4, (const char*)samples + (size_t)accepted[i] * n * 4,
                       (jn - i) * (size_t)n * 4, kind, nullptr);
    i = jn;
  }
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(stream)");
  rc = process_batch_impl(c, d_pad, dtype, n_frames, 256, nullptr, nullptr, nullptr, (float2*)c->s_rx_mag.p, nullptr);
  if (rc) return rc;
  c->h_rx_mag.resize(n_frames);
  e = hipMemcpy(c->h_rx_mag.data(), c->s_rx_mag.p, n_frames * sizeof(float2), hipMemcpyDeviceToHost);  // syncs stream 0
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(mag_max)");

  RxReplay rx{c->h_rx_mag.data(), n_frames, n, 0};
  uchirp::MainLoop<RxReplay> loop(n, c->cfg.snr_threshold);
  size_t nt = 0, ntext = 0;
  auto put = [&](char ch) {
    if (ntext + 1 < text_cap) text[ntext++] = ch;
  };
  for (size_t i = 0; i < na; i++) {
    rx.block = i;
    const uchirp::loop_event le = loop.step(rx, put);
    if (trace && nt < trace_cap) {
      uc_rx_event& ev = trace[nt];
      ev.block = accepted[i];
      ev.sync_position = le.sync_position;
      ev.state_before = (uint8_t)le.state_before;
      ev.state_after = (uint8_t)le.state_after;
      ev.bit = (int8_t)le.bit;
      ev.reserved = 0;
      ev.snr_up = le.snr_up;
      ev.snr_down = le.snr_down;
    }
    nt++;
  }
  text[ntext] = '\0';
  if (n_trace) *n_trace = nt;
  return (int)ntext;
}

extern "C" int uc_receive_stream(uc_ctx* c, const void* samples, int dtype, size_t n_samples, char* text,
                                 size_t text_cap, uc_rx_event* trace, size_t trace_cap, size_t* n_trace) {
  return uc_receive_stream_isr(c, samples, dtype, n_samples, nullptr, text, text_cap, trace, trace_cap, n_trace);
}

// ---------------------------------------------------------------------------
// uc_receive_streams[_next]: the same receiver for MANY streams at once, recorded or live.  Per call: (busy mask only:
// accept + pack) -> ONE launch of the band kernel's ROWS build over the 8 FIFO offsets every accepted block adds (it reads
// the caller's buffer as it lies) -> main()'s switch replayed on the device, one wave or one lane per stream
// (csrc/uc_rx_kernel.hip: include/uchirp_mainloop.hpp compiled for the device) -> (live: the stream's newest block is kept).
// ---------------------------------------------------------------------------
// live streams: what n_streams receivers carry from one call to the next, all of it on the device
struct uc_rx_state {
  uc_ctx* c = nullptr;
  int device = 0;
  size_t n_streams = 0;
  uint32_t* d_last = nullptr;   // [2][n_streams][n] words: the newest ACCEPTED block of every stream -- the part of the FIFO the
                                // next block's new offsets still read; zeros at power-on (fifo_queue, main.c:94).  Two
                                // halves: a call reads half *d_parity and fills the other one (the band kernel stores the
                                // block on its way through; no copy kernel, no frame reads what another writes)
  unsigned int* d_parity = nullptr;  // which half is current; flipped by the replay kernel of every call
  float2* d_carry = nullptr;    // [n_streams][9]: (up, down) mag_max of the 9 FIFO offsets that survive the ISR's shift
                                // (main.c:662): offsets n .. 2 n of the FIFO become 0 .. n of the next one; zeros at power-on
  uint32_t* d_loop = nullptr;   // [n_streams][rx_loop_words()]: main()'s locals (main.c:314-339) + blocks offered so far
  uint32_t* d_need = nullptr;   // [n_streams]: what the switch can still look at of every stream's NEXT block (which of its 8 new
                                // FIFO offsets, and the DOWN statistics: main.c:447-453; uc_rx.hpp); written by every replay
  uint32_t* d_hist = nullptr;   // [n_streams][4] PDM words: the DFSDM's sinc^5 history of every microphone (UC_DTYPE_PDM chunks)
  RxScratch rx;                 // scratch of the call in flight
  uint64_t blocks_seen = 0;     // host mirror of the block count (the overflow check only; a replayed graph does not bump it)
  int dtype = -1;               // of the words in d_last (the first call decides)
  // uc_rx_state_keep_previous: the caller keeps the chunk of every call alive and unchanged until the NEXT call on the state has
  // completed, so "the block in front" of a call's first block is read where the previous call's samples lie -- nothing is
  // saved into d_last.  kept = the last block of stream 0 of the previous call's chunk (device memory of the caller),
  // kept_pitch elements from stream to stream; nullptr: the FIFO's newest block is in d_last (power-on, after a busy-masked
  // call, after a call on host memory).
  bool keep = false;
  const void* kept = nullptr;
  size_t kept_pitch = 0;
};

// Everything uc_receive_streams[_next] refuses for its ARGUMENTS, and nothing else: no HIP call that enqueues, no allocation.
// receive_streams_impl runs it first; uc_group_receive_streams[_next] runs it for EVERY local device before it touches a stream
// (uc_group.cpp: a refused group call has started nothing -- ADVICE r5).  0, or the negative code the call would return.
int uc::receive_streams_check(uc_ctx* c, uc_rx_state* st, bool live, const void* samples, int dtype, size_t n_streams, size_t n_samples,
                              size_t stream_stride_elems, const char* text, size_t text_cap, size_t trace_cap) {
  if (!c || !text || text_cap == 0) return fail(-EINVAL, "uc_receive_streams: NULL argument");
  if (live && (!st || st->c != c)) return fail(-EINVAL, "uc_receive_streams_next: the state belongs to another context");
  if (c->cfg.variant != UC_RX_REAL && c->cfg.variant != UC_SYNC_CPLX)
    return fail(-ENOTSUP, "uc_receive_streams: variant %d has no up/down state machine", (int)c->cfg.variant);
  if (dtype != UC_DTYPE_I32 && dtype != UC_DTYPE_F32 && dtype != UC_DTYPE_PDM)
    return fail(-EINVAL, "uc_receive_streams: bad dtype %d", dtype);
  if (n_streams == 0) return 0;
  const uint32_t n = c->cfg.n;
  if (stream_stride_elems == 0) stream_stride_elems = n_samples;
  if (stream_stride_elems < n_samples) return fail(-EINVAL, "uc_receive_streams: streams overlap (stride %zu < %zu samples)",
                                                   stream_stride_elems, n_samples);
  const size_t nb = n_samples / n;
  if (nb >= ((size_t)1 << 31) / 8 || text_cap >= ((size_t)1 << 31) || trace_cap >= ((size_t)1 << 31))
    return fail(-EINVAL, "uc_receive_streams: stream too long");
  if (st) {
    if (n_samples % n != 0) return fail(-EINVAL, "uc_receive_streams_next: %zu samples are not whole blocks of %u", n_samples, n);
    if (st->dtype >= 0 && st->dtype != dtype && st->blocks_seen)
      return fail(-EINVAL, "uc_receive_streams_next: the streams began as dtype %d", st->dtype);
    if (st->blocks_seen + nb >= ((uint64_t)1 << 32)) return fail(-EOVERFLOW, "uc_receive_streams_next: 2^32 blocks per stream");
  }
  if (nb) {
    if (!samples) return fail(-EINVAL, "uc_receive_streams: samples is NULL");
    if (n_streams * nb * (size_t)(n / 256) >= ((size_t)1 << 31))
      return fail(-EINVAL, "uc_receive_streams: %zu new FIFO offsets in one call (at most 2^31 - 1)", n_streams * nb * (size_t)(n / 256));
    // UC_DTYPE_PDM on device memory: the DFSDM kernel's alignment rules (a host buffer is staged into aligned scratch)
    if (dtype == UC_DTYPE_PDM && is_device_ptr(samples) &&
        (((uintptr_t)samples & 15u) != 0 || (n_streams > 1 && (stream_stride_elems & 3u) != 0)))
      return fail(-EINVAL, "uc_receive_streams: UC_DTYPE_PDM device buffers must be 16-byte aligned, the stride a multiple of 4 words");
  }
  return 0;
}

static int receive_streams_impl(uc_ctx* c, uc_rx_state* st, const void* samples, int dtype, size_t n_streams, size_t n_samples,
                                size_t stream_stride_elems, const uint8_t* busy, char* text, size_t text_cap,
                                uint32_t* n_text, uc_rx_event* trace, size_t trace_cap, uint32_t* n_trace,
                                void* hip_stream) {
  if (const int rc = uc::receive_streams_check(c, st, st != nullptr, samples, dtype, n_streams, n_samples, stream_stride_elems, text,
                                               text_cap, trace_cap))
    return rc;
  if (n_streams == 0) return 0;
  const uint32_t n = c->cfg.n;
  const uint32_t per_block = n / 256;  // new FIFO offsets per accepted block
  const bool pdm = dtype == UC_DTYPE_PDM;
  const int dtype_in = dtype;
  const size_t stream_stride_in = stream_stride_elems ? stream_stride_elems : n_samples;
  if (stream_stride_elems == 0) stream_stride_elems = n_samples;
  const size_t nb = n_samples / n;
  if (trace && trace_cap == 0) trace = nullptr;
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;
  bool host_out = false;
  RxScratch& sc = st ? st->rx : c->rx;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  const bool capturing = stream && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
  // Capture is the live form's: a uc_rx_state owns its scratch.  The context's scratch serves every state-less call, on any
  // stream, ordered by an event a capture cannot carry -- a captured state-less call would share it with eager calls and
  // replays without any ordering
  if (!st && capturing)
    return fail(-ENOTSUP, "uc_receive_streams: the stream is being captured -- only uc_receive_streams_next (a uc_rx_state owns "
                          "its scratch) can be captured into a hipGraph");
  const CaptureNoAlloc no_alloc(capturing);
  if (!st && !capturing) {
    // the context's scratch serves one call at a time: a call on another stream than the last one waits, ON THE DEVICE, for
    // that one's kernels (a live state has scratch of its own and needs none of this)
    if (c->rx_used && c->rx_stream != stream) {
      const RelaxedCapture relaxed;
      e = hipStreamWaitEvent(stream, c->rx_ev, 0);
      if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent(receiver scratch)");
    }
  }

  // outputs: straight into device buffers of the caller, else through staging
  char* d_text = text;
  if (!is_device_ptr(text)) {
    if (int rc = sc.text.ensure(n_streams * text_cap)) return rc;
    d_text = (char*)sc.text.p;
    host_out = true;
  }
  uint32_t* d_ntext = n_text;
  if (n_text && !is_device_ptr(n_text)) {
    if (int rc = sc.ntext.ensure(n_streams * sizeof(uint32_t))) return rc;
    d_ntext = (uint32_t*)sc.ntext.p;
    host_out = true;
  }
  uc_rx_event* d_trace = trace;
  if (trace && !is_device_ptr(trace)) {
    if (int rc = sc.trace.ensure(n_streams * trace_cap * sizeof(uc_rx_event))) return rc;
    d_trace = (uc_rx_event*)sc.trace.p;
    host_out = true;
  }
  uint32_t* d_ntrace = n_trace;
  if (n_trace && !is_device_ptr(n_trace)) {
    if (int rc = sc.ntrace.ensure(n_streams * sizeof(uint32_t))) return rc;
    d_ntrace = (uint32_t*)sc.ntrace.p;
    host_out = true;
  }
  if (nb == 0) {  // nothing to process: empty texts, zero counts
    e = hipMemsetAsync(d_text, 0, n_streams * text_cap, stream);
    if (e == hipSuccess && d_ntext) e = hipMemsetAsync(d_ntext, 0, n_streams * sizeof(uint32_t), stream);
    if (e == hipSuccess && d_ntrace) e = hipMemsetAsync(d_ntrace, 0, n_streams * sizeof(uint32_t), stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(empty streams)");
  }

  if (nb) {
    const size_t n_frames = n_streams * nb * per_block;
    // (every scratch buffer of the call is sized before its first launch: a call that cannot be served -- out of memory, or a
    // capture that would have to allocate -- has enqueued nothing)
    if (int rc = sc.rec.ensure(n_frames * sizeof(float2))) return rc;
    if (pdm)
      if (int rc = sc.pcm.ensure(n_streams * nb * (size_t)n * 4)) return rc;
    if (busy) {
      int rc = sc.acc.ensure(n_streams * nb * sizeof(uint32_t));
      if (!rc) rc = sc.na.ensure(n_streams * sizeof(uint32_t));
      if (!rc) rc = sc.pad.ensure(n_streams * nb * (size_t)n * 4);
      if (!rc && !is_device_ptr(busy)) rc = sc.busy.ensure(n_streams * nb);
      if (rc) return rc;
    }
    // inputs
    const void* d_in = samples;
    if (!is_device_ptr(samples)) {
      const size_t span = (n_streams - 1) * stream_stride_elems + nb * (size_t)n;
      if (int rc = sc.in.ensure(span * 4)) return rc;
      e = hipMemcpyAsync(sc.in.p, samples, span * 4, hipMemcpyHostToDevice, stream);
      if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(streams)");
      d_in = sc.in.p;
    }
    if (pdm) {
      // The chain starts at the microphones' bit streams: the DFSDM (receiver/Src/dfsdm.c:59-61,69,78) turns every 32 PDM
      // bits into one word of `buf[]` -- whole blocks here, the filter history of every stream carried in the state (or,
      // for streams that start with this call, the bit pattern of a silent microphone) -- and the ISR sees int32 words.
      // The peripheral filters whether or not the ISR later drops the block: the busy mask applies behind it.
      uint32_t* d_hist = st ? st->d_hist : nullptr;
      if (!st) {
        if (int rc = sc.hist.ensure(n_streams * 16)) return rc;
        d_hist = (uint32_t*)sc.hist.p;
        e = hipMemsetD32Async((hipDeviceptr_t)d_hist, (int)UC_PDM_SILENCE, n_streams * 4, stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemsetD32Async(pdm history)");
      }
      if (int rc = sc.pcm.ensure(n_streams * nb * (size_t)n * 4)) return rc;
      if (int rc = sinc5_streams_launch(c, (const uint32_t*)d_in, n_streams, nb * (size_t)n, stream_stride_elems, d_hist, true,
                                        (int32_t*)sc.pcm.p, nb * (size_t)n, stream))
        return rc;
      d_in = sc.pcm.p;
      stream_stride_elems = nb * (size_t)n;
      dtype = UC_DTYPE_I32;
    }
    // The ISR (main.c:659-668) appends a block only when the main loop has consumed the previous one; a block that arrives
    // while it is busy is DROPPED, the FIFO is not shifted.  The FIFO therefore only ever holds ACCEPTED blocks: with a busy
    // mask they are first laid out one behind the other; without one the caller's buffer is read as it lies.
    const void* rows = d_in;
    size_t row_pitch = stream_stride_elems;
    const uint32_t* d_acc = nullptr;
    const uint32_t* d_na = nullptr;
    // uc_rx_state_keep_previous holds for chunks the caller owns on the device and calls that accept every block; a call on
    // host memory (staged by the library), from PDM bits (the DFSDM words are the library's) or with a busy mask (the newest
    // ACCEPTED block differs by stream) hands the FIFO's newest block to the state as ever
    const bool keep_next = st && st->keep && !busy && !pdm && d_in == samples;
    if (st && st->kept && busy) {
      // a busy-masked call behind kept chunks: a stream whose blocks are all dropped keeps its FIFO, so the kept blocks go
      // into the state's current half first -- from here on this call is an ordinary one
      const bool al16 = (((uintptr_t)st->kept | (uintptr_t)st->d_last) & 15u) == 0 && (st->kept_pitch & 3u) == 0 && (n & 3u) == 0;
      const int lrc = uc::launch_rx_keep(st->kept, st->kept_pitch, n, n_streams, st->d_last, st->d_parity, al16, stream);
      if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx keep kernel launch");
      st->kept = nullptr;
      st->kept_pitch = 0;
    }
    if (busy) {
      const uint8_t* d_busy = busy;
      if (!is_device_ptr(busy)) {
        if (int rc = sc.busy.ensure(n_streams * nb)) return rc;
        e = hipMemcpyAsync(sc.busy.p, busy, n_streams * nb, hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(busy)");
        d_busy = (const uint8_t*)sc.busy.p;
      }
      int rc = sc.acc.ensure(n_streams * nb * sizeof(uint32_t));
      if (!rc) rc = sc.na.ensure(n_streams * sizeof(uint32_t));
      if (!rc) rc = sc.pad.ensure(n_streams * nb * (size_t)n * 4);
      if (rc) return rc;
      int lrc = uc::launch_rx_accept(d_busy, n_streams, (uint32_t)nb, (uint32_t*)sc.acc.p, (uint32_t*)sc.na.p, stream);
      if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx accept kernel launch");
      d_acc = (const uint32_t*)sc.acc.p;
      d_na = (const uint32_t*)sc.na.p;
      const bool al16 = (((uintptr_t)d_in | (uintptr_t)sc.pad.p) & 15u) == 0 && (stream_stride_elems & 3u) == 0 && (n & 3u) == 0;
      lrc = uc::launch_rx_pack(d_in, stream_stride_elems, n, (uint32_t)nb, n_streams, d_acc, d_na, sc.pad.p, nb * (size_t)n, al16,
                               stream);
      if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx pack kernel launch");
      rows = sc.pad.p;
      row_pitch = nb * (size_t)n;
    }
    if (int rc = sc.rec.ensure(n_frames * sizeof(float2))) return rc;
    // dsp() at the 8 FIFO offsets every accepted block ADDS (the other 9 of its FIFO were evaluated when the block before
    // it arrived, main.c:662): frame (s, k, m) = the last n - 256 m samples of the block in front of the stream's k-th
    // block followed by the first 256 m of that block -- in front of block 0: the newest block of the previous call (a
    // live state) or zeros (power-on).  (Rows of a busy-masked call beyond na[s] hold stale words: evaluated, never read.)
    {
      uc::BandParams bp;
      memset(&bp, 0, sizeof(bp));
      bp.frames = rows;
      bp.n_frames = n_frames;
      bp.magmax = (float2*)sc.rec.p;
      bp.prev = st ? (const void*)st->d_last : (const void*)c->d_zero_block;
      bp.prev_pitch = st ? (size_t)n : 0;
      bp.prev_half = st ? n_streams * (size_t)n : 0;
      bp.parity = st ? st->d_parity : nullptr;
      bp.save = (st && !busy) ? 1u : 0u;  // (with a busy mask the last ACCEPTED block differs by stream: launch_rx_last)
      bp.save_to = st ? (void*)st->d_last : nullptr;
      bp.save_half = bp.prev_half;
      if (st && st->kept) {
        // the caller has kept the previous chunk (uc_rx_state_keep_previous): the block in front is read where it lies
        bp.prev = st->kept;
        bp.prev_pitch = st->kept_pitch;
        bp.prev_half = 0;
      }
      if (keep_next) bp.save = 0u;  // ... and this call's chunk will be kept for the next one: nothing to hand over
      // acquisition evaluates 4 positions a block, the UP reference only: a stream that is IDLE when its ONE new block arrives
      // gets the 3 or 5 transforms the switch can still look at (SYNC_CPLX: of the UP reference only) instead of 8
      bp.need = (st && nb == 1) ? st->d_need : nullptr;
      bp.poison = c->rx_poison ? 1u : 0u;
      bp.row_pitch = row_pitch;
      bp.row_blocks = (uint32_t)nb;
      uc::rows_divisor((uint32_t)nb, &bp.div_magic, &bp.div_shift);
      if (int rc = band_launch(c, bp, dtype, stream)) return rc;
    }
    uc::RxParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.rec = (const float2*)sc.rec.p;
    rp.rec_pitch = nb * per_block;
    rp.carry = st ? st->d_carry : (const float2*)c->d_zero_block;
    rp.carry_pitch = st ? per_block + 1 : 0;
    rp.carry_out = st ? st->d_carry : nullptr;
    rp.n_streams = n_streams;
    rp.n = n;
    rp.nb = (uint32_t)nb;
    rp.snr_threshold = c->cfg.snr_threshold;
    rp.acc = d_acc;
    rp.na = d_na;
    rp.text = d_text;
    rp.text_cap = (uint32_t)text_cap;
    rp.n_text = d_ntext;
    rp.trace = d_trace;
    rp.trace_cap = (uint32_t)trace_cap;
    rp.n_trace = d_ntrace;
    rp.loop_state = st ? st->d_loop : nullptr;
    rp.parity = st ? st->d_parity : nullptr;
    rp.need = st ? st->d_need : nullptr;
    rp.need_force = c->rx_need_force;
    int lrc;
    if (st && busy) {
      // what the next call's new offsets still read of this one: every stream's newest ACCEPTED block
      const bool al16 = (((uintptr_t)rows | (uintptr_t)st->d_last) & 15u) == 0 && (row_pitch & 3u) == 0 && (n & 3u) == 0;
      lrc = uc::launch_rx_last(rows, row_pitch, d_na, (uint32_t)nb, n, n_streams, st->d_last, st->d_parity, al16, stream);
      if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx last-block kernel launch");
    }
    lrc = uc::launch_rx_replay(rp, stream);
    if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx replay kernel launch");
    if (st) {
      st->blocks_seen += nb;
      st->dtype = dtype_in;
      st->kept = keep_next ? (const void*)((const char*)samples + (nb - 1) * (size_t)n * 4) : nullptr;
      st->kept_pitch = keep_next ? stream_stride_in : 0;
    }
  }
  if (host_out) {
    if (d_text != text) e = hipMemcpyAsync(text, d_text, n_streams * text_cap, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && n_text && d_ntext != n_text)
      e = hipMemcpyAsync(n_text, d_ntext, n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && trace && d_trace != trace)
      e = hipMemcpyAsync(trace, d_trace, n_streams * trace_cap * sizeof(uc_rx_event), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess && n_trace && d_ntrace != n_trace)
      e = hipMemcpyAsync(n_trace, d_ntrace, n_streams * sizeof(uint32_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return hip_fail(e, "uc_receive_streams: copy back");
  }
  if (!st && !capturing) {
    const RelaxedCapture relaxed;
    if (hipEventRecord(c->rx_ev, stream) == hipSuccess) {
      c->rx_used = true;
      c->rx_stream = stream;
    } else {
      // (a capture elsewhere on this thread forbids the call): nothing can be waited for later -- drain now
      (void)hipGetLastError();
      (void)hipStreamSynchronize(stream);
      c->rx_used = false;
    }
  }
  return 0;
}

extern "C" int uc_receive_streams(uc_ctx* c, const void* samples, int dtype, size_t n_streams, size_t n_samples,
                                  size_t stream_stride_elems, const uint8_t* busy, char* text, size_t text_cap,
                                  uint32_t* n_text, uc_rx_event* trace, size_t trace_cap, uint32_t* n_trace,
                                  void* hip_stream) {
  return receive_streams_impl(c, nullptr, samples, dtype, n_streams, n_samples, stream_stride_elems, busy, text, text_cap, n_text,
                              trace, trace_cap, n_trace, hip_stream);
}

// ---- live streams: the same call, chunk after chunk ----------------------------------------------------------------
extern "C" void uc_rx_state_destroy(uc_rx_state* st) {
  if (!st) return;
  (void)hipSetDevice(st->device);  // (not through st->c: a state may outlive its context by mistake; its memory is its own)
  if (st->d_last) (void)hipFree(st->d_last);
  if (st->d_carry) (void)hipFree(st->d_carry);
  if (st->d_loop) (void)hipFree(st->d_loop);
  if (st->d_parity) (void)hipFree(st->d_parity);
  if (st->d_hist) (void)hipFree(st->d_hist);
  if (st->d_need) (void)hipFree(st->d_need);
  st->rx.release();
  delete st;
}

extern "C" size_t uc_rx_state_streams(const uc_rx_state* st) { return st ? st->n_streams : 0; }

extern "C" int uc_rx_state_keep_previous(uc_rx_state* st, int on) {
  if (!st) return fail(-EINVAL, "uc_rx_state_keep_previous: NULL state");
  st->keep = on != 0;  // (switched off: the NEXT call still reads the kept chunk in front of its own, and saves its own)
  return 0;
}

extern "C" int uc_rx_state_reset(uc_rx_state* st, void* hip_stream) {
  if (!st) return fail(-EINVAL, "uc_rx_state_reset: NULL state");
  uc_ctx* c = st->c;
  hipError_t e = hipSetDevice(st->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  hipStream_t stream = (hipStream_t)hip_stream;
  const uint32_t n = c->cfg.n;
  e = hipMemsetAsync(st->d_last, 0, 2 * st->n_streams * (size_t)n * 4, stream);
  if (e == hipSuccess) e = hipMemsetAsync(st->d_parity, 0, sizeof(unsigned int), stream);
  if (e == hipSuccess)  // (power-on: IDLE, turn 0 -- the word the replay would have left)
    e = hipMemsetD32Async((hipDeviceptr_t)st->d_need, (int)uc::need_word(UC_STATE_IDLE, 0, 0), st->n_streams, stream);
  if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)st->d_hist, (int)UC_PDM_SILENCE, st->n_streams * 4, stream);
  if (e == hipSuccess) e = hipMemsetAsync(st->d_carry, 0, st->n_streams * (size_t)(n / 256 + 1) * sizeof(float2), stream);
  if (e != hipSuccess) return hip_fail(e, "hipMemsetAsync(FIFO state)");
  const int lrc = uc::launch_rx_state_init(st->d_loop, st->n_streams, n, c->cfg.snr_threshold, stream);
  if (lrc != (int)hipSuccess) return hip_fail((hipError_t)lrc, "rx state init kernel launch");
  st->blocks_seen = 0;
  st->dtype = -1;
  st->kept = nullptr;
  st->kept_pitch = 0;
  return 0;
}

extern "C" int uc_rx_state_create(uc_ctx* c, size_t n_streams, uc_rx_state** out) {
  if (!c || !out) return fail(-EINVAL, "uc_rx_state_create: NULL argument");
  *out = nullptr;
  if (c->cfg.variant != UC_RX_REAL && c->cfg.variant != UC_SYNC_CPLX)
    return fail(-ENOTSUP, "uc_rx_state_create: variant %d has no up/down state machine", (int)c->cfg.variant);
  if (n_streams == 0) return fail(-EINVAL, "uc_rx_state_create: no streams");
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) return hip_fail(e, "hipSetDevice");
  uc_rx_state* st = new (std::nothrow) uc_rx_state();
  if (!st) return fail(-ENOMEM, "uc_rx_state_create: out of memory");
  st->c = c;
  st->device = c->device;
  st->n_streams = n_streams;
  const uint32_t n = c->cfg.n;
  e = hipMalloc((void**)&st->d_last, 2 * n_streams * (size_t)n * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_parity, 256);
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_need, n_streams * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_hist, n_streams * 16);
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_carry, n_streams * (size_t)(n / 256 + 1) * sizeof(float2));
  if (e == hipSuccess) e = hipMalloc((void**)&st->d_loop, n_streams * (size_t)uc::rx_loop_words() * 4);
  if (e != hipSuccess) {
    uc_rx_state_destroy(st);
    return hip_fail(e, "hipMalloc(rx state)");
  }
  int rc = uc_rx_state_reset(st, nullptr);
  if (!rc) {
    e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) rc = hip_fail(e, "hipStreamSynchronize");
  }
  if (rc) {
    uc_rx_state_destroy(st);
    return rc;
  }
  *out = st;
  return 0;
}

extern "C" int uc_receive_streams_next(uc_ctx* c, uc_rx_state* st, const void* samples, int dtype, size_t n_samples,
                                       size_t stream_stride_elems, const uint8_t* busy, char* text, size_t text_cap,
                                       uint32_t* n_text, uc_rx_event* trace, size_t trace_cap, uint32_t* n_trace,
                                       void* hip_stream) {
  if (!st || !c || st->c != c) return fail(-EINVAL, "uc_receive_streams_next: the state belongs to another context");
  return receive_streams_impl(c, st, samples, dtype, st->n_streams, n_samples, stream_stride_elems, busy, text, text_cap, n_text,
                              trace, trace_cap, n_trace, hip_stream);
}
