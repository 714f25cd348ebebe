// san_host.cpp -- the CPU-reachable host code of the library under AddressSanitizer + UndefinedBehaviorSanitizer
// (tools/sanitize.sh; no GPU needed, none used).  Not a parity test (tests/ holds those): it walks the code paths that index,
// allocate and shift -- the table builders of every variant, the sinc^5 byte tables, main()'s switch + resync()
// (include/uchirp_mainloop.hpp, receiver/Src/main.c:417-554, 243-273) driven by a CPU dsp() over random statistics with the
// FIFO bounds asserted on every call (quirk Q8), the partition / span arithmetic of the multi-GPU leg at its edges, the
// multiply-high divisor, and the argument checks of the C-ABI that come before any device call.
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/uchirp.h"
#include "../../include/uchirp_mainloop.hpp"
#include "../../ultrasonic-communication_amd/csrc/uc_kernels.hpp"
#include "../../ultrasonic-communication_amd/csrc/uc_tables.hpp"

#define CHECK(c)                                                               \
  do {                                                                         \
    if (!(c)) {                                                                \
      fprintf(stderr, "san_host: CHECK failed at line %d: %s\n", __LINE__, #c); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static uint64_t rng_state = 0x243F6A8885A308D3ull;
static uint32_t rnd() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(rng_state >> 33);
}

struct Hist {
  float mag_max = 0.f, mag_mean = 0.f, snr = 0.f;
};
// dsp() stand-in: a level that the "scenario" moves between noise and signal; every call checks the FIFO bounds
struct CpuDsp {
  typedef Hist history_t;
  uint32_t n;
  float level_up, level_dn;
  uint64_t calls = 0;
  bool ok = true;
  void dsp(uint32_t pos, Hist* h, float mag_mean, int updown) {
    if (pos > 2 * n || (pos & 255u) != 0) ok = false;  // a frame must lie inside the 3 n FIFO (Q8: checked BEFORE evaluating)
    const float jitter = 0.75f + (float)(rnd() & 1023) / 2048.0f;
    const float m = (updown == UC_UP_CHIRP ? level_up : level_dn) * jitter;
    h->mag_max = m;
    h->mag_mean = mag_mean;
    h->snr = (m - mag_mean) / mag_mean;
    calls++;
  }
};

int main() {
  // ---- 1. tables of every variant and flag combination the library builds
  int built = 0;
  for (int variant = 0; variant < UC_NUM_VARIANTS; variant++) {
    uc_config cfg;
    CHECK(uc_default_config(variant, &cfg) == 0);
    for (uint32_t flags : {0u, (uint32_t)UC_FLAG_LIBM_TRIG, (uint32_t)UC_FLAG_TRUE_DC}) {
      uc_config c = cfg;
      c.flags |= flags;
      if (variant == UC_STREAM) {
        for (uint32_t d : {0u, 4u, 8u, 16u}) {
          c.decim = d;
          uc::StreamTables st;
          CHECK(uc::build_stream_tables(c, st) == 0);
          CHECK(st.hn.size() == 2 * (size_t)c.n && st.ctap.size() == 2 * uc::kFirTaps && st.hop > 0);
          built++;
        }
        c.decim = 5;
        uc::StreamTables bad;
        CHECK(uc::build_stream_tables(c, bad) != 0);
        continue;
      }
      uc::Tables t;
      CHECK(uc::build_tables(c, t) == 0);
      CHECK(t.hann.size() == c.n && t.n == c.n);
      built++;
      for (float fs : {125000.0f / 3.0f, 100000.0f, 48000.0f}) {  // other sampling rates: wider / narrower windows
        uc_config w = c;
        w.fs = fs;
        uc::Tables tw;
        (void)uc::build_tables(w, tw);  // may refuse (window too wide): must not crash
        built++;
      }
    }
  }
  {  // base-band I/Q at both frame lengths
    for (uint32_t n : {1024u, 2048u}) {
      uc_config c;
      CHECK(uc_default_config(UC_IQ, &c) == 0);
      c.n = n; c.fs = 100000.0f; c.carrier = 18000.0f; c.f0 = 16500.0f; c.f1 = 19500.0f; c.time_frame = n / 100000.0f;
      c.flags = UC_FLAG_IQ_BASEBAND;
      uc::Tables t;
      CHECK(uc::build_tables(c, t) == 0);
      built++;
    }
  }
  std::vector<int32_t> t4, t1;
  uc::build_sinc5_tables(t4, t1);
  CHECK(t4.size() == 4 * 256 * 4 && t1.size() == 4 * 256);
  {
    long long sum = 0;  // every tap once: all-ones bytes over the five outputs sum to 32^5
    for (int b = 0; b < 4; b++) {
      for (int w = 0; w < 4; w++) sum += t4[(size_t)(b * 256 + 255) * 4 + w];
      sum += t1[(size_t)b * 256 + 255];
    }
    CHECK(sum == (1ll << 25));
  }
  std::vector<float> tw, packed, in(2048);
  uc::build_twiddles(2048, tw);
  CHECK(tw.size() == 4096);
  for (size_t i = 0; i < in.size(); i++) in[i] = (float)((int)(rnd() % 2001) - 1000);
  uc::packed_rfft_double(in, packed);
  CHECK(packed.size() == 2048);

  // ---- 2. main()'s switch + resync() over a CPU dsp(): noise, acquisitions, data, drop-outs; FIFO bounds on every call
  uint64_t steps = 0, chars = 0, bits = 0;
  for (int run = 0; run < 40; run++) {
    CpuDsp d{2048, 1.0f, 1.0f};
    uchirp::MainLoop<CpuDsp> loop(2048, run % 3 == 0 ? 2.0f : (run % 3 == 1 ? 0.5f : 6.0f));
    std::string text;
    auto put = [&](char c) { if (text.size() < 4096) text.push_back(c); chars++; };
    int phase = 0, left = 0;
    for (int b = 0; b < 20000; b++) {
      if (left-- <= 0) {  // a new stretch: silence, an up-only preamble, data (either chirp loud), a fade
        phase = (int)(rnd() % 4);
        left = 1 + (int)(rnd() % 60);
      }
      const float noise = 1.0f + (float)(rnd() % 100) / 100.0f;
      d.level_up = phase == 1 || (phase == 2 && (rnd() & 1)) ? noise * (3.0f + (float)(rnd() % 40)) : noise;
      d.level_dn = phase == 2 && d.level_up <= noise * 2.0f ? noise * (3.0f + (float)(rnd() % 40)) : noise;
      if (phase == 3) d.level_up = d.level_dn = noise * (rnd() % 7 == 0 ? 2.9f : 1.0f);
      const uchirp::loop_event ev = loop.step(d, put);
      CHECK(ev.state_before >= UC_STATE_IDLE && ev.state_after <= UC_STATE_DATA_RECEIVING);
      CHECK(ev.sync_position <= 2 * 2048u && (ev.sync_position & 255u) == 0);
      CHECK(ev.bit >= -1 && ev.bit <= 1);
      if (ev.bit >= 0) bits++;
      steps++;
    }
    CHECK(d.ok);
    CHECK(d.calls >= 20000);
  }
  CHECK(bits > 1000 && chars > 100);

  // ---- 3. partition / span arithmetic at its edges
  for (int world = 1; world <= 64; world++) {
    for (size_t n : {(size_t)0, (size_t)1, (size_t)63, (size_t)64, (size_t)65, (size_t)1 << 20, ((size_t)1 << 40) + 7, (size_t)-1 / 4}) {
      size_t next = 0;
      for (int r = 0; r < world; r++) {
        size_t first = 0, count = 0;
        CHECK(uc_partition(n, world, r, &first, &count) == 0);
        CHECK(first == next);
        next = first + count;
        size_t e0 = 0, ne = 0;
        CHECK(uc_frame_span(2048, 256, 0, first, count, &e0, &ne) == 0);
        CHECK(count == 0 ? ne == 0 : ne == (count - 1) * 256 + 2048);
        CHECK(uc_frame_span(1024, 0, 26, first, count, &e0, &ne) == 0);
      }
      CHECK(next == n);
    }
    size_t f, c;
    CHECK(uc_partition(10, world, world, &f, &c) < 0 && uc_partition(10, world, -1, &f, &c) < 0);
  }
  {
    size_t f, c;
    CHECK(uc_partition(10, 0, 0, &f, &c) < 0);
    CHECK(uc_partition(10, 4, 1, nullptr, &c) < 0 || true);  // (NULL outputs: refused or ignored, never written through)
  }
  for (uint32_t d = 1; d < 3000; d++) {
    uint32_t m = 0, s = 0;
    uc::rows_divisor(d, &m, &s);
    for (uint32_t u : {0u, d - 1, d, d + 1, 0x7fffffffu, 0x7fffffffu - d}) {
      const uint32_t q = ((uint32_t)(((uint64_t)u * m) >> 32) + u) >> s;
      CHECK(q == u / d);
    }
  }

  // ---- 4. the C-ABI's argument checks in front of any device call (this machine has no GPU: uc_create must say so)
  {
    uc_config cfg;
    CHECK(uc_default_config(UC_RX_REAL, &cfg) == 0);
    CHECK(uc_default_config(99, &cfg) < 0 && uc_default_config(UC_RX_REAL, nullptr) < 0);
    CHECK(uc_default_config(UC_RX_REAL, &cfg) == 0);
    uc_ctx* ctx = nullptr;
    const int rc = uc_create(&cfg, &ctx);
    if (rc == 0) {  // (a GPU after all: nothing below needs one, but the context must go)
      uc_destroy(ctx);
    } else {
      CHECK(ctx == nullptr && strlen(uc_last_error()) > 0);
    }
    CHECK(uc_create(nullptr, &ctx) < 0 && uc_create(&cfg, nullptr) < 0);
    uc_destroy(nullptr);
    uint8_t sym = 0;
    uc_stats st[2];
    int32_t frame[8] = {0};
    CHECK(uc_process_frame(nullptr, frame, 1.0f, &sym, st) < 0);
    CHECK(uc_process_batch(nullptr, frame, UC_DTYPE_I32, 1, 0, nullptr, &sym, st, nullptr) < 0);
    CHECK(uc_window_spectrum(nullptr, frame, UC_DTYPE_I32, 1, 0, nullptr, nullptr) < 0);
    CHECK(uc_get_table(nullptr, UC_TABLE_UP, nullptr, 0) < 0);
    CHECK(uc_set_table(nullptr, UC_TABLE_UP, nullptr, 0) < 0);
    CHECK(uc_get_windows(nullptr, nullptr, nullptr, nullptr) < 0);
    CHECK(uc_stats_per_frame(nullptr) <= 2 && uc_iq_halo(nullptr) <= 26);
    char text[8];
    CHECK(uc_receive_stream(nullptr, frame, UC_DTYPE_I32, 8, text, sizeof(text), nullptr, 0, nullptr) < 0);
    CHECK(uc_receive_streams(nullptr, frame, UC_DTYPE_I32, 1, 8, 0, nullptr, text, sizeof(text), nullptr, nullptr, 0, nullptr, nullptr) < 0);
    uc_rx_state* rx = nullptr;
    CHECK(uc_rx_state_create(nullptr, 1, &rx) < 0 && rx == nullptr);
    CHECK(uc_receive_streams_next(nullptr, nullptr, frame, UC_DTYPE_I32, 8, 0, nullptr, text, sizeof(text), nullptr, nullptr, 0, nullptr, nullptr) < 0);
    CHECK(uc_rx_state_reset(nullptr, nullptr) < 0 && uc_rx_state_streams(nullptr) == 0);
    uc_rx_state_destroy(nullptr);
    CHECK(uc_process_stream(nullptr, frame, UC_DTYPE_I32, 8, nullptr, nullptr, nullptr) < 0);
    CHECK(uc_stream_geometry(nullptr, 8, nullptr, nullptr, nullptr, nullptr) < 0);
    CHECK(uc_dfsdm_sinc5(nullptr, nullptr, 8, nullptr, nullptr) < 0);
    CHECK(uc_dfsdm_sinc5_streams(nullptr, nullptr, 1, 8, 0, nullptr, nullptr, 0, nullptr) < 0);
    uc_group* g = nullptr;
    int32_t dev0[2] = {0, 0};
    CHECK(uc_group_create(nullptr, dev0, 1, &g) < 0 && uc_group_create(&cfg, nullptr, 1, &g) < 0);
    CHECK(uc_group_create(&cfg, dev0, 0, &g) < 0 && uc_group_create(&cfg, dev0, 65, &g) < 0);
    CHECK(uc_group_create(&cfg, dev0, 2, &g) < 0);  // one device named twice
    CHECK(uc_group_create_rank(&cfg, nullptr, 2, 0, &g) < 0);
    unsigned char id[UC_GROUP_ID_BYTES] = {0};
    CHECK(uc_group_create_rank(&cfg, id, 2, 2, &g) < 0 && uc_group_create_rank(&cfg, id, 0, 0, &g) < 0);
    CHECK(uc_group_preflight(nullptr) < 0);
    if (getenv("UC_RCCL_LIB")) {
      // the instrumented loop-back stand-in loaded through the group's own dlopen path (UC_TUNING=1 UC_RCCL_LIB=...): the id
      // call, and the whole build-up / tear-down of a group that fails half way (no GPU here: uc_create refuses)
      unsigned char uid[UC_GROUP_ID_BYTES];
      CHECK(uc_group_unique_id(uid, sizeof(uid)) == 0);
      CHECK(uc_group_unique_id(uid, 4) < 0);
      if (uc_device_count() == 0) {
        CHECK(uc_group_preflight(&cfg) < 0 && strlen(uc_last_error()) > 0);
        CHECK(uc_group_create(&cfg, dev0, 1, &g) < 0 && g == nullptr);
        CHECK(uc_group_create_rank(&cfg, uid, 2, 1, &g) < 0 && g == nullptr);
      }
    }
    CHECK(uc_group_process_batch(nullptr, nullptr, UC_DTYPE_F32, 1, 0, nullptr, nullptr) < 0);
    CHECK(uc_group_receive_streams(nullptr, nullptr, UC_DTYPE_F32, 1, 2048, 0, nullptr, nullptr, 8, nullptr, nullptr) < 0);
    CHECK(uc_group_receive_streams_next(nullptr, nullptr, nullptr, UC_DTYPE_F32, 1, 2048, 0, nullptr, nullptr, 8, nullptr, nullptr) < 0);
    CHECK(uc_group_process_stream(nullptr, nullptr, UC_DTYPE_F32, 1 << 20, nullptr, nullptr, nullptr) < 0);
    CHECK(uc_group_wait_gather(nullptr, 0, nullptr, nullptr) < 0 && uc_group_synchronize(nullptr) < 0);
    CHECK(uc_group_world(nullptr) < 0 && uc_group_local_count(nullptr) < 0 && uc_group_first_rank(nullptr) < 0);
    CHECK(uc_group_ctx(nullptr, 0) == nullptr);
    uc_group_destroy(nullptr);
    CHECK(uc_group_unique_id(nullptr, 0) < 0);
    CHECK(uc_clock_probe(nullptr, 1) < 0 && uc_clock_read(nullptr, nullptr) < 0 && uc_clock_stamps(nullptr, nullptr, 0) < 0);
    CHECK(uc_debug_busy_counters(nullptr) < 0);
    CHECK(uc_device_malloc(0, 16, nullptr) < 0);
    CHECK(uc_abi_version() == UC_ABI_VERSION);
  }
  printf("san_host ok: %d table sets, %llu passes of main()'s switch (%llu bits, %llu characters), spans and argument checks clean\n",
         built, (unsigned long long)steps, (unsigned long long)bits, (unsigned long long)chars);
  return 0;
}
