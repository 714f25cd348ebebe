// uc_api_core.cpp -- the C-ABI of include/uchirp.h on top of the gfx950 kernels: errors, contexts and their tables, the hand-out
// counters of the dynamically dealt launches, the band launch, uc_process_batch / uc_process_frame / uc_window_spectrum.
// (The receivers: uc_api_rx.cpp.  UC_STREAM: uc_api_stream.cpp.  The DFSDM: uc_api_cic.cpp.  The clock probe: uc_api_clock.cpp.)
// No CPU compute path exists here: without a usable HIP device uc_create fails.
#include "uc_api_internal.hpp"

using namespace uc_api;

namespace {
thread_local std::string g_err;
}

thread_local bool uc_api::g_capturing = false;

int uc_api::fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int uc_api::hip_fail(hipError_t e, const char* what) {
  return fail(-EIO, "%s: %s", what, hipGetErrorString(e));
}

bool uc_api::is_device_ptr(const void* p) {
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

namespace uc {
void set_error(const char* msg) { g_err = msg ? msg : ""; }  // (uc_group.cpp reports through the same uc_last_error())
}


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
int uc_api::upload(void** dst, const void* src, size_t bytes) {
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
    if (const char* g = getenv("UC_RX_STEP_MIN")) c->rx_step_min = atol(g);
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

int uc_api::take_work_counter(uc_ctx* c, hipStream_t stream, unsigned int** out, int* slot) {
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
int uc_api::work_counter_launched(uc_ctx* c, hipStream_t stream, int slot) {
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
int uc_api::clock_buffer(uc_ctx* c, size_t grid, int waves_per_wg, hipStream_t stream, unsigned long long** out) {
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
int uc_api::band_launch(uc_ctx* c, uc::BandParams& p, int dtype, hipStream_t stream) {
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

int uc_process_batch(uc_ctx* c, const void* frames, int dtype, size_t n_frames, size_t stride_elems,
                     const float* mag_mean, uint8_t* symbols, uc_stats* stats, void* hip_stream) {
  return process_batch_impl(c, frames, dtype, n_frames, stride_elems, mag_mean, symbols, stats, nullptr, hip_stream);
}

// d_magmax: device, (up, down) mag_max per frame, nullable (internal: uc_receive_stream)
// mapped  : every pointer is device-accessible as it stands (internal: the pinned, mapped frame slot of uc_process_frame)
// d_spectrum: device, the window bins of every frame (internal: uc_window_spectrum; band variants only)
int uc_api::process_batch_impl(uc_ctx* c, const void* frames, int dtype, size_t n_frames, size_t stride_elems,
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
